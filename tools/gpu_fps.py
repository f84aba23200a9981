#!/usr/bin/env python3
"""K1 (temporal kernel) time per output frame for the three ring sizes: 30 fps (8 taps), 60 fps (15 -> 16), 120 fps (30 -> 32)."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
N = 120
t, r = synth_video_pair(N, 2160, 3840, device="cuda")
for fps in (30, 60, 120):
    m = fv.fvvdp(display_name="standard_4k"); m.timing = True
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); q, st = m.predict(t, r, frames_per_second=fps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    print("fps %3d: predict %.2f ms, K1 %.1f us per output frame (%d launches), bands %.1f us/frame" % (
        fps, dt * 1e3, ms[0] / (3 * N) * 1e3, cnt[0], sum(ms[1:9]) / (3 * N) * 1e3), flush=True)
