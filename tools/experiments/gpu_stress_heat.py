#!/usr/bin/env python3
"""Randomised sweep of the 'raw' difference maps: HIP path against the oracle.  usage: gpu_stress_heat.py [cases] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from oracle import fvvdp_oracle as orc
from fovvideovdp_amd.synth import synth_video_pair
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst, worst_m, fails = (0.0, None), (0.0, None), 0
for case in range(n_cases):
    H, W = int(rng.integers(17, 150)), int(rng.integers(17, 260))
    fps = int(rng.choice([0, 24, 30, 60, 120]))
    N = 1 if fps == 0 else int(rng.integers(2, 9))
    pad = str(rng.choice(["replicate", "circular", "pingpong"]))
    disp = str(rng.choice(["standard_4k", "standard_fhd", "standard_hdr_pq", "standard_hmd"]))
    fov = bool(rng.integers(0, 3) == 0)
    t, r = synth_video_pair(N, H, W, pair=int(rng.integers(0, 50)))
    fix = np.array([W * 0.3, H * 0.6]) if fov else None
    desc = f"{W}x{H}x{N} fps={fps} pad={pad} {disp} fov={fov}"
    try:
        q, st = fv.fvvdp(display_name=disp, temp_padding=pad, foveated=fov, heatmap="raw").predict(t, r, frames_per_second=fps, fixation_point=fix)
        oq, ost = orc.Oracle(disp, temp_padding=pad, foveated=fov, heatmap="raw").predict(t.numpy(), r.numpy(), "BCFHW", fps, fix)
    except Exception as e:
        print("raise", desc, str(e)[:80]); continue
    h, g = st["heatmap"].float().numpy().astype(np.float64), ost["heatmap"].astype(np.float64)
    rel = float(np.max(np.abs(h - g) / np.maximum(np.abs(g), 2e-3))) * 1024
    mean = float(np.mean(np.abs(h - g)))
    if rel > worst[0]: worst = (rel, desc)
    if mean > worst_m[0]: worst_m = (mean, desc)
    if rel > 3.0 or abs(float(q) - float(oq)) > 1e-4:
        print("FAIL", desc, "ulp %.2f dJOD %.2e" % (rel, abs(float(q) - float(oq)))); fails += 1
print("heat cases", n_cases, "fails", fails, "| worst %.2f fp16 ulp (%s) | worst mean abs %.2e (%s)" % (worst[0], worst[1], worst_m[0], worst_m[1]))
