#!/usr/bin/env python3
"""Parity sweep over extreme frame shapes (very wide, very tall, tiny, sizes around the strip / tile boundaries of the
kernels) against the oracle, image and short video."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from oracle import fvvdp_oracle as orc
from fovvideovdp_amd.synth import synth_video_pair
shapes = [(16, 4096), (17, 3001), (2048, 16), (1999, 17), (16, 16), (16, 17), (17, 16), (31, 33), (64, 120), (64, 121), (64, 124),
          (64, 125), (64, 240), (64, 241), (64, 244), (64, 245), (33, 247), (33, 248), (33, 249), (33, 256), (33, 257), (120, 64),
          (121, 63), (255, 255), (256, 256), (257, 257), (40, 1024), (41, 1023), (300, 500)]
worst, fails = (0.0, None), 0
for (H, W) in shapes:
    for (N, fps) in ((1, 0), (5, 30)):
        t, r = synth_video_pair(N, H, W, pair=3)
        tn, rn = t.numpy(), r.numpy()
        desc = f"{W}x{H}x{N}"
        try:
            m = fv.fvvdp(display_name="standard_4k")
            q, st = m.predict(tn, rn, frames_per_second=fps)
            hip_err = None
        except Exception as e:
            hip_err = str(e)[:70]
        try:
            oq, ost = orc.Oracle("standard_4k").predict(tn, rn, frames_per_second=fps)
            orc_err = None
        except Exception as e:
            orc_err = str(e)[:70]
        if hip_err or orc_err:
            same = bool(hip_err) == bool(orc_err)
            print("raise" if same else "FAIL (one side raised)", desc, "| HIP:", hip_err, "| oracle:", orc_err)
            fails += 0 if same else 1
            continue
        dq = abs(float(q) - float(oq))
        a, b = st["Q_per_ch"].astype(np.float64), ost["Q_per_ch"].astype(np.float64)
        rel = float(np.max(np.abs(a - b) / (np.abs(b) + 1e-5 * b.max() + 1e-12)))
        if dq > worst[0]: worst = (dq, desc)
        if dq > 5e-4 or rel > 1e-2:
            print("FAIL", desc, "dJOD %.2e relQ %.2e" % (dq, rel)); fails += 1
print("shapes", len(shapes), "fails", fails, "| worst dJOD %.2e (%s)" % worst)
