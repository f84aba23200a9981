#!/usr/bin/env python3
"""Measured error of the HIP path against golden g10 (every other display model of the reference, plain and foveated)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fovvideovdp_amd as fv
from test_oracle_golden import G10_DISPLAYS, g10_inputs
z = np.load(os.path.join(ROOT, "tests", "golden", "g10_displays.npz"))
w = {False: [0, 0], True: [0, 0]}
for disp in G10_DISPLAYS:
    t, r, gaze = g10_inputs(disp)
    for fov in (False, True):
        tag = disp + ("_fov" if fov else "")
        m = fv.fvvdp(display_name=disp, foveated=fov)
        q, st = m.predict(t, r, frames_per_second=30, fixation_point=gaze.numpy() if fov else None)
        gq = z[tag + "_Q"].astype(np.float64); qq = st["Q_per_ch"].astype(np.float64)
        dj = abs(float(q) - float(z[tag + "_jod"])); dq = float(np.max(np.abs(qq - gq) / (np.abs(gq) + 1e-3 * np.max(gq))))
        w[fov][0] = max(w[fov][0], dj); w[fov][1] = max(w[fov][1], dq)
        print("%-26s dJOD %.2e  dQ %.2e" % (tag, dj, dq))
print("worst plain: dJOD %.2e dQ %.2e | foveated: dJOD %.2e dQ %.2e" % (w[False][0], w[False][1], w[True][0], w[True][1]))
