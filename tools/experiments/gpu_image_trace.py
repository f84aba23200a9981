#!/usr/bin/env python3
"""30 predict() calls on a 512x512 image (for rocprofv3 --kernel-trace --memory-copy-trace: what fills the GPU time line of a small call)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
test, ref = synth_video_pair(1, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_fhd")
for _ in range(30):
    q, st = m.predict(test, ref)
torch.cuda.synchronize()
print(float(q))
