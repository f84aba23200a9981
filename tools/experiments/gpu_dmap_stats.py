#!/usr/bin/env python3
"""Measured error of the per-band difference maps D against the reference's captures (goldens g2): per map the largest absolute
error relative to the map's maximum and the quantiles of the relative error the test bounds (tests/test_gpu_parity.py::stage_check)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fovvideovdp_amd as fv
import test_gpu_parity as tp
tp._D_STATS = []
for (H, W, N, fps) in [(135, 240, 10, 30), (68, 121, 12, 60)]:
    tp.test_stages_from_golden_R.__wrapped__(fv, H, W, N, fps) if hasattr(tp.test_stages_from_golden_R, "__wrapped__") else tp.test_stages_from_golden_R(fv, H, W, N, fps)
st = tp._D_STATS
print("maps %d" % len(st))
a = np.array([s[2:] for s in st])
print("worst over maps: abs/max %.2e | rel q99 %.2e | rel q99.9 %.2e | rel max %.2e | rel mean %.2e" % tuple(a.max(axis=0)))
big = [s for s in st if s[1] >= 4096]
b = np.array([s[2:] for s in big])
print("maps of >= 4096 pixels (%d): abs/max %.2e | rel q99 %.2e | rel q99.9 %.2e | rel max %.2e | rel mean %.2e" % ((len(big),) + tuple(b.max(axis=0))))
for s in sorted(st, key=lambda s: -s[2])[:5]:
    print("  %s px %d abs/max %.2e q99 %.2e q99.9 %.2e max %.2e mean %.2e" % s)
