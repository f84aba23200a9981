#!/usr/bin/env python3
"""Where the host time of one predict() call goes (4K x60 resident pair): cProfile of 20 calls + wall per call."""
import cProfile, pstats, os, sys, time, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = int(os.environ.get("HH", 2160)), int(os.environ.get("WW", 3840)), 60
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
for _ in range(3): m.predict(test, ref, frames_per_second=30)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): m.predict(test, ref, frames_per_second=30)
torch.cuda.synchronize()
print("wall per call %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
# host time to enqueue everything (no sync): sync=False
t0 = time.perf_counter()
for _ in range(20): m.predict(test, ref, frames_per_second=30, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host enqueue per call (sync=False) %.3f ms; with drain %.3f ms per call" % ((t1 - t0) / 20 * 1e3, (time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): m.predict(test, ref, frames_per_second=30)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
