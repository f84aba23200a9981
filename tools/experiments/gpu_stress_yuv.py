#!/usr/bin/env python3
"""Randomised parity sweep of the YUV ingest path against the oracle (sizes, bit depths, chroma formats, colour spaces,
frame rates).  usage: gpu_stress_yuv.py [cases] [seed]
RESIZE=1 in the environment: every case also draws a full-screen resize (method and target size; the per-frame path with
fvvdp_yuv_frame_resized) -- a separate population, so that the seeds of the sweeps on record keep their cases."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from oracle import fvvdp_oracle as orc
from fovvideovdp_amd.synth import synth_yuv_pair
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst, fails = (0.0, None), 0
for case in range(n_cases):
    css = str(rng.choice(["420", "444"]))
    H, W = int(rng.integers(9, 70)) * 2, int(rng.integers(9, 120)) * 2
    bd = int(rng.choice([8, 10, 12]))
    fps = int(rng.choice([24, 30, 50, 60, 120]))
    N = int(rng.integers(2, 12))
    cs = str(rng.choice(["bt709", "bt2020nc"]))
    disp = str(rng.choice(["standard_4k", "standard_fhd", "standard_hdr_pq"]))
    ty, ry = synth_yuv_pair(N, H, W, bit_depth=bd, chroma_ss=css, pair=int(rng.integers(0, 9)))
    desc = f"{W}x{H}x{N} {css} {bd}bit {cs} fps={fps} {disp}"
    rs = {}
    if os.environ.get("RESIZE"):
        rs = dict(full_screen_resize=str(rng.choice(["bilinear", "bicubic", "nearest", "area"])),
                  resize_resolution=(int(rng.integers(20, 260)), int(rng.integers(20, 150))))
        desc += " -> %s %dx%d" % (rs["full_screen_resize"], *rs["resize_resolution"])
    m = fv.fvvdp(display_name=disp)
    vs = fv.fvvdp_video_source_yuv_frames(ty, ry, fps, W, H, bit_depth=bd, chroma_ss=css, color_space=cs,
                                          display_photometry=m.display_photometry, **rs)
    q, st = m.predict_video_source(vs)
    tn = ty.numpy() if bd == 8 else ty.numpy().astype(np.uint16)
    rn = ry.numpy() if bd == 8 else ry.numpy().astype(np.uint16)
    o = orc.Oracle(disp, color_space="BT.2020" if cs == "bt2020nc" else "sRGB")
    oq, ost = o.predict_yuv(tn, rn, fps, W, H, bit_depth=bd, chroma_ss=css, color_space=cs, **rs)
    dq = abs(float(q) - float(oq))
    if dq > worst[0]: worst = (dq, desc)
    if dq > 5e-4:
        print("FAIL", desc, "dJOD %.2e" % dq); fails += 1
print("cases", n_cases, "fails", fails, "| worst dJOD %.2e (%s)" % worst)
