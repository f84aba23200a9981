#!/usr/bin/env python3
"""Timing of the secondary input/output variants of predict() at 4K (u16 / fp32 sources, HDR displays, heat maps)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair

H, W, N = 2160, 3840, int(sys.argv[1]) if len(sys.argv) > 1 else 30
t8, r8 = synth_video_pair(N, H, W, device="cuda")


def run(name, test, ref, display, fps=30, **kw):
    m = fv.fvvdp(display_name=display, **{k: v for k, v in kw.items() if k in ("heatmap", "foveated")})
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, st = m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-34s %7.2f ms  %8.0f Mpix/s  JOD %.5f" % (name, best * 1e3, 2 * W * H * N / best / 1e6, float(q)), flush=True)
    del m


run("uint8 sRGB (baseline)", t8, r8, "standard_4k")
t16 = (t8.to(torch.int32) * 257).to(torch.int16)
r16 = (r8.to(torch.int32) * 257).to(torch.int16)
run("uint16 (int16 carrier) sRGB", t16, r16, "standard_4k")
del t16, r16
tf = t8.to(torch.float32) / 255.0
rf = r8.to(torch.float32) / 255.0
run("fp32 [0,1] sRGB", tf, rf, "standard_4k")
run("fp32 [0,1] PQ HDR display", tf, rf, "standard_hdr_pq")
tl = tf * 400.0 + 0.1
rl = rf * 400.0 + 0.1
del tf, rf
run("fp32 linear cd/m2 HDR display", tl, rl, "standard_hdr_linear")
del tl, rl
run("uint8 + heatmap=threshold", t8, r8, "standard_4k", heatmap="threshold")
run("uint8 foveated", t8, r8, "standard_4k", foveated=True)
run("uint8 60 fps (fl=15)", t8, r8, "standard_4k", fps=60)
run("uint8 120 fps (fl=30)", t8, r8, "standard_4k", fps=120)
