"""CPU-baseline timing helper for bench.py (test infrastructure, like everything under oracle/).

One output frame of a clip per process: `python -m oracle.cpu_bench <test.npy> <ref.npy> <first> <fl> <fps> <display>
<start_at>` loads frames [first, first+fl) (the temporal window of output frame first+fl-1), sets the oracle up, waits
until the wall-clock time `start_at` so that all workers compute concurrently, evaluates the frame with the numpy
oracle and prints `<seconds> <end_time>`.  bench.py starts the workers as plain subprocesses with a hard timeout: no
multiprocessing machinery, no torch and no HIP runtime in the workers."""
import os
import subprocess
import sys
import time

import numpy as np


def _worker(argv):
    test_npy, ref_npy, first, fl, fps, display, start_at = argv[0], argv[1], int(argv[2]), int(argv[3]), float(argv[4]), argv[5], float(argv[6])
    from oracle import fvvdp_oracle as orc
    t = np.ascontiguousarray(np.load(test_npy, mmap_mode="r")[:, :, first:first + fl])
    r = np.ascontiguousarray(np.load(ref_npy, mmap_mode="r")[:, :, first:first + fl])
    gaze = os.environ.get("ORACLE_FOVEATED_GAZE")                # "x,y": foveated mode with a fixed gaze (BASELINE configs[3])
    fix = [float(v) for v in gaze.split(",")] if gaze else None
    o = orc.Oracle(display, foveated=fix is not None)
    o.predict(t[:, :, :1, :32, :32], r[:, :, :1, :32, :32], frames_per_second=0, fixation_point=[16, 16] if fix else None)   # touch the LUTs
    while time.time() < start_at:
        time.sleep(0.005)
    t0 = time.perf_counter()
    o.predict(t, r, frames_per_second=fps, frames=[fl - 1], fixation_point=fix)
    dt = time.perf_counter() - t0
    print("%.6f %.6f" % (dt, time.time()), flush=True)


def timed_frames(test, ref, fps, display, fl, n_procs, tmp_dir, timeout=180.0, gaze=None):
    """Evaluates n_procs output frames of the clip (frames fl-1, fl, ...; wrapping around when there are more processes than
    full windows in the clip: every frame costs the same), one per subprocess, all at the same time.
    Returns (wall seconds from the common start to the last finish, per-frame seconds) or raises RuntimeError."""
    n_win = max(1, test.shape[2] - fl + 1)             # distinct full temporal windows in the clip
    need = fl - 1 + min(n_procs, n_win)
    tp, rp = os.path.join(tmp_dir, "cpu_t.npy"), os.path.join(tmp_dir, "cpu_r.npy")
    np.save(tp, np.ascontiguousarray(test[:, :, :need]))
    np.save(rp, np.ascontiguousarray(ref[:, :, :need]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=root)
    env.pop("ORACLE_FOVEATED_GAZE", None)
    if gaze is not None:                                     # foveated mode, one fixed gaze position (x, y) for every worker
        env["ORACLE_FOVEATED_GAZE"] = "%g,%g" % (float(gaze[0]), float(gaze[1]))
    start_at = time.time() + 6.0 + 0.05 * n_procs           # interpreter + numpy start-up, frame loading
    procs = []
    try:
        for k in range(n_procs):
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.cpu_bench", tp, rp, str(k % n_win), str(fl), str(fps),
                                           display, "%.6f" % start_at], cwd=root, env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.DEVNULL, text=True))
        deadline = time.time() + timeout
        secs, ends = [], []
        for p in procs:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            if p.returncode != 0:
                raise RuntimeError("cpu baseline worker failed")
            a, b = out.split()
            secs.append(float(a))
            ends.append(float(b))
        late = max(e - s for e, s in zip(ends, secs)) - start_at    # a worker that was not ready at start_at
        if late > 0.5:
            raise RuntimeError("cpu baseline workers did not start together")
        return max(ends) - start_at, secs
    except (subprocess.TimeoutExpired, ValueError) as e:
        raise RuntimeError("cpu baseline workers timed out") from e
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for f in (tp, rp):
            if os.path.exists(f):
                os.remove(f)


if __name__ == "__main__":
    _worker(sys.argv[1:])
