#!/usr/bin/env python3
"""Per-call latency of small inputs (BASELINE configs[0]/[1]-like): host time per predict() and GPU kernel time.
usage: gpu_small_latency.py [HxW[xN] ...]   (N > 1: video at 30 fps)"""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair


def run(H, W, N, reps=200):
    test, ref = synth_video_pair(max(N, 1), H, W, device="cuda")
    m = fv.fvvdp(display_name="standard_fhd")
    kw = dict(frames_per_second=30) if N > 1 else {}
    if N <= 1:
        test, ref = test[:, :, :1], ref[:, :, :1]
    for _ in range(5):
        q, st = m.predict(test, ref, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        q, st = m.predict(test, ref, **kw)
    torch.cuda.synchronize()
    t_sync = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    outs = []
    for _ in range(reps):
        outs.append(m.predict(test, ref, sync=False, **kw))
    t_host = (time.perf_counter() - t0) / reps
    torch.cuda.synchronize()
    t_async = (time.perf_counter() - t0) / reps
    for q, st in outs:
        m.finish(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        m.predict(test, ref, sync=False, **kw)
    e1.record()
    torch.cuda.synchronize()
    print("%dx%dx%d: predict() %.1f us/call synchronous | sync=False: host %.1f us/call, completed %.1f us/call, GPU-side %.1f us/call | JOD %.4f"
          % (H, W, N, t_sync * 1e6, t_host * 1e6, t_async * 1e6, e0.elapsed_time(e1) * 1e3 / reps, float(q)))


if __name__ == "__main__":
    cases = sys.argv[1:] or ["512x512", "1080x1920", "512x512x12", "1080x1920x12"]
    for c in cases:
        p = [int(v) for v in c.split("x")]
        run(p[0], p[1], p[2] if len(p) > 2 else 1)
