#!/usr/bin/env python3
"""Measured error of the HIP path against golden g11 (photometry / geometry objects built by the caller, 1.5-9 ppd)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair, synth_gaze
from test_oracle_golden import G11_CASES
z = np.load(os.path.join(ROOT, "tests", "golden", "g11_custom_display_objects.npz"))
N, H, W = 10, 90, 160
test, ref = synth_video_pair(N, H, W); gaze = synth_gaze(N, H, W)
w = {False: [0, 0], True: [0, 0]}
for tag, pcls, pkw, gkw in G11_CASES:
    for fov in (False, True):
        t = tag + ("_fov" if fov else "")
        m = fv.fvvdp(display_name="standard_4k", display_photometry=getattr(fv, pcls)(**pkw),
                     display_geometry=fv.fvvdp_display_geometry((W, H), **gkw), foveated=fov)
        q, st = m.predict(test, ref, frames_per_second=30, fixation_point=gaze.numpy() if fov else None)
        gq = z[t + "_Q"].astype(np.float64); qq = st["Q_per_ch"].astype(np.float64)
        dj = abs(float(q) - float(z[t + "_jod"])); dq = float(np.max(np.abs(qq - gq) / (np.abs(gq) + 1e-3 * np.max(gq))))
        w[fov][0] = max(w[fov][0], dj); w[fov][1] = max(w[fov][1], dq)
        print("%-22s bands %d  dJOD %.2e  dQ %.2e" % (t, qq.shape[0], dj, dq))
print("worst plain: dJOD %.2e dQ %.2e | foveated: dJOD %.2e dQ %.2e" % (w[False][0], w[False][1], w[True][0], w[True][1]))
