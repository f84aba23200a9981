#!/usr/bin/env python3
"""Per-kernel timing probe (HIP events inside the library) for A/B builds: FVVDP_LIB=<variant.so> tools/experiments/gpu_speed.py"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
sizes = [(2160, 3840, 60)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for (H, W, N) in sizes:
    test, ref = synth_video_pair(N, H, W, device="cuda")
    m = fv.fvvdp(display_name="standard_4k" if H >= 2160 else "standard_fhd")
    m.timing = True
    best = 1e9
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, st = m.predict(test, ref, frames_per_second=30)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        if it == 1:
            ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
            nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    nb = m._ctx.key[2]
    per = [ms[i] / max(cnt[i], 1) / N * 1e3 for i in range(nb + 2)]
    print("%s %dx%dx%d best %.2f ms %.0f Mpix/s JOD %.6f | us/frame: temporal %.1f  bands %s  fin %.2f | bands total %.1f" % (
        os.path.basename(nat.LIB_PATH), W, H, N, best * 1e3, 2 * W * H * N / best / 1e6, float(q), per[0],
        " ".join("%.2f" % x for x in per[1:nb + 1]), per[nb + 1], sum(per[1:nb + 2])), flush=True)
