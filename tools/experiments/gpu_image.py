#!/usr/bin/env python3
"""Latency of still-image predict() calls (per call; device-resident and host numpy inputs)."""
import cProfile, pstats, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
for (H, W, disp) in [(1080, 1920, "standard_fhd"), (2160, 3840, "standard_4k")]:
    t, r = synth_video_pair(1, H, W, device="cuda")
    ti, ri = t[0, :, 0].contiguous(), r[0, :, 0].contiguous()          # CHW on the device
    tn, rn = ti.permute(1, 2, 0).cpu().numpy().copy(), ri.permute(1, 2, 0).cpu().numpy().copy()   # HWC host
    m = fv.fvvdp(display_name=disp)
    for name, (a, b, do) in {"device CHW": (ti, ri, "CHW"), "host HWC numpy": (tn, rn, "HWC")}.items():
        for _ in range(3):
            q, _ = m.predict(a, b, dim_order=do)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            q, _ = m.predict(a, b, dim_order=do)
            float(q)
        dt = (time.perf_counter() - t0) / n
        print("%dx%d image, %s: %.3f ms per call (%.0f Mpix/s) JOD %.5f" % (W, H, name, dt * 1e3, 2 * W * H / dt / 1e6, float(q)), flush=True)
    if H == 2160:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(20):
            q, _ = m.predict(ti, ri, dim_order="CHW"); float(q)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
