#!/usr/bin/env python3
"""Foveated stage 2 only (target for rocprofv3 --pmc): 4K x30 frames, standard_hdr_pq, moving gaze."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair, synth_gaze
N, H, W = 30, 2160, 3840
test, ref = synth_video_pair(N, H, W, device="cuda")
gaze = synth_gaze(N, H, W).numpy()
m = fv.fvvdp(display_name="standard_hdr_pq", foveated=True)
for _ in range(3):
    q, st = m.predict(test, ref, frames_per_second=30, fixation_point=gaze)
torch.cuda.synchronize()
print("JOD", float(q))
