#!/usr/bin/env python3
"""Benchmark of the FovVideoVDP hot path on MI355X.

A "step" is one full pass of the hot path over one synthetic 3840x2160 x 60-frame video pair per GPU
(BASELINE.json configs[2]: uint8 RGB, standard_4k, 30 fps -> 8-tap temporal filter, foveated off), inputs already
resident in HBM: unpack + sRGB display model + luminance + temporal filtering + pyramid + CSF + masking + pooling
+ pooling and JOD regression.  value = Mpixels/s (test+ref) = 2*W*H*N*steps*n_gpus / wall seconds.

  python bench.py [--gpus N --steps K --warmup W]           (N>1: launched by torch.distributed.run)

One JSON line on rank 0; `roofline` is measured live with HIP events inside the library around the dominant
kernel (fused pyramid level 0), `cpu_baseline` times the numpy oracle on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def level_sizes(W, H, n_bands):
    out = []
    w, h = W, H
    for _ in range(n_bands + 1):
        out.append((w, h))
        w, h = (w + 1) // 2, (h + 1) // 2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--fps", type=int, default=30)
    ap.add_argument("--display", default="standard_4k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--cpu-procs", type=int, default=8, help="processes of the CPU baseline (1 = time the oracle in-process)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to "
                    "dry-run the multi-rank code path on a single-GPU box)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()      # dry run: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    assert args.gpus == world, "--gpus must equal the number of launched ranks"

    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    from fovvideovdp_amd.sharding import gather_pair_results

    W, H, N, fps = args.width, args.height, args.frames, args.fps
    test, ref = synth_video_pair(N, H, W, device=dev, pair=rank)     # each rank owns its own pair (pair sharding)
    m = fv.fvvdp(display_name=args.display, device=dev)

    def step():
        q, stats = m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
        if world > 1:
            # the one collective of the path: every rank's Q_per_ch lands in its own slot of a zero buffer
            allq = gather_pair_results(torch.from_numpy(stats["Q_per_ch"]).to(dev), rank, world)
            return m.do_pooling_and_jods(allq, None).tolist()        # all pairs pooled in one batched call
        return [float(q)]

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        jods = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        jods = step()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    mpix = 2.0 * W * H * N * args.steps * world / dt / 1e6

    # ---- roofline of the dominant kernel (fused pyramid level 0), HIP events on the kernels' own stream ----
    roof = None
    extra = {}
    if rank == 0:
        from fovvideovdp_amd import _native as nat
        import ctypes as C
        nat.check(nat.lib().fvvdp_ctx_timing_enable(m._ctx.handle, 1))      # same context (same HBM scratch) as the timed steps
        nk = 16 + 2
        ms = (C.c_float * nk)()
        cnt = (C.c_int32 * nk)()
        reps = 5
        nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, nk, 1))
        for _ in range(reps):
            m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
        nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, nk, 1))
        n_bands = m._ctx.key[2]
        batch = m._ctx.key[4]
        sizes = level_sizes(W, H, n_bands)
        P = 4
        # algorithmic bytes of one level-0 launch: read level 0 once, write level 1 once, all P planes fp32
        launches0 = max(int(cnt[1]), 1)
        frames_per_launch = N * reps / launches0
        b0 = 4.0 * P * (sizes[0][0] * sizes[0][1] + sizes[1][0] * sizes[1][1]) * frames_per_launch
        t0_ms = ms[1] / launches0
        ach = b0 / (t0_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "band_kernel<4> level 0 (pyramid+CSF+masking+pooling)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": None, "avg_launch_ms": round(t0_ms, 4), "bytes_per_launch": int(b0),
                "frames_per_launch": frames_per_launch}
        # HBM bytes per launch from the PMC counters cannot be collected from inside this process; they come from the
        # committed rocprofv3 --pmc passes of the same kernel and launch shape (tools/pmc_level0.py, profiles/)
        pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_level0.json")) \
            if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        if pmc and (W, H, int(frames_per_launch)) == (3840, 2160, 60):
            with open(os.path.join(ROOT, "profiles", pmc[-1])) as f:
                pj = json.load(f)
            roof["traffic"] = int(pj["traffic_bytes"])
            roof["traffic_source"] = "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)" % pmc[-1]
        # graded pass = all band levels (B_alg of SURVEY section 8(d))
        b_all = sum(4.0 * P * (sizes[i][0] * sizes[i][1] + sizes[i + 1][0] * sizes[i + 1][1]) for i in range(n_bands)) * N * reps
        t_all = sum(ms[1 + i] for i in range(n_bands)) + ms[1 + n_bands]
        extra["graded_pass"] = {"levels_ms_per_frame": [round(ms[1 + i] / (N * reps), 5) for i in range(n_bands)],
                                "finalize_ms_per_frame": round(ms[1 + n_bands] / (N * reps), 5),
                                "temporal_ms_per_frame": round(ms[0] / (N * reps), 5),
                                "hbm_frac_all_levels": round(b_all / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "us_per_frame_all_levels": round(t_all * 1e3 / (N * reps), 2)}
        extra["batch_frames"] = batch
        nat.check(nat.lib().fvvdp_ctx_timing_enable(m._ctx.handle, 0))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import fvvdp_oracle as orc
        tc, rc = test.cpu().numpy(), ref.cpu().numpy()
        fl = fl_guard(fps) + 1
        done = False
        if args.cpu_procs > 1:
            # the numpy oracle on several host cores: one output frame (with its own temporal window) per process,
            # all running concurrently (oracle/cpu_bench.py; plain subprocesses with a hard timeout)
            import tempfile
            from oracle import cpu_bench
            try:
                with tempfile.TemporaryDirectory() as d:
                    wall, per = cpu_bench.timed_frames(tc, rc, fps, args.display, fl, args.cpu_procs, d)
                cpu = {"value": round(2.0 * W * H * args.cpu_procs / wall / 1e6, 3), "unit": "Mpixels/s (test+ref)",
                       "cores": args.cpu_procs, "kind": "port",
                       "sample": "%d output frames (frames %d..%d, each incl. its %d-frame temporal window) of the same %dx%d "
                                 "pair, one per process, numpy fp32 oracle, %d processes concurrently (%.1f - %.1f s per frame, "
                                 "%.1f s wall); host has %d cores" % (args.cpu_procs, fl - 1, fl - 2 + args.cpu_procs, fl, W, H,
                                                                     args.cpu_procs, min(per), max(per), wall, os.cpu_count())}
                done = True
            except Exception as e:                      # never let the baseline leg break the benchmark line
                sys.stderr.write("cpu baseline: parallel run failed (%s), timing a single process instead\n" % e)
        if not done:
            nf = max(1, args.cpu_frames)
            frames = list(range(fl - 1, fl - 1 + nf))
            o = orc.Oracle(args.display)
            tcpu = time.perf_counter()
            o.predict(tc, rc, frames_per_second=fps, frames=frames)
            tcpu = time.perf_counter() - tcpu
            cpu = {"value": round(2.0 * W * H * nf / tcpu / 1e6, 3), "unit": "Mpixels/s (test+ref)", "cores": 1,
                   "kind": "port", "sample": "%d output frames (frames %d..%d, incl. their %d-frame temporal window) of the same "
                   "%dx%d pair, numpy fp32 oracle, single thread; host has %d cores" % (nf, frames[0], frames[-1], fl, W, H, os.cpu_count())}

    # second half of the metric: |JOD - JOD of the reference| for rank 0's pair, from the committed golden
    # (tests/golden/g3_synth_uhd_60f.npz = the reference's own torch-CPU run on this synthetic pair, tools/gen_golden.py g3)
    jod_delta = None
    gpath = os.path.join(ROOT, "tests", "golden", "g3_synth_uhd_60f.npz")
    if rank == 0 and (W, H, N, fps, args.display) == (3840, 2160, 60, 30, "standard_4k") and os.path.exists(gpath):
        jod_delta = abs(float(jods[0]) - float(np.load(gpath)["jod"]))

    if rank == 0:
        out = {"metric": "Mpixels/s (test+ref) at 4Kx60f; JOD delta vs reference", "value": round(mpix, 1),
               "unit": "Mpixels/s (test+ref)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%dx%d x%d-frame synthetic uint8 RGB video pair per GPU, %s, %d fps, foveated=off "
                                      "(BASELINE.json configs[2]); pairs sharded one per GPU" % (W, H, N, args.display, fps),
                          "parallelism": "pair-sharded x%d, one all-reduce of Q_per_ch" % world},
               "jod": [round(j, 6) for j in jods],
               "jod_delta_vs_reference": None if jod_delta is None else float("%.3g" % jod_delta), "roofline": roof, "cpu_baseline": cpu}
        out.update(extra)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def fl_guard(fps):
    return int(np.ceil(250.0 / (1000.0 / fps))) - 1


if __name__ == "__main__":
    main()
