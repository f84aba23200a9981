#!/usr/bin/env python3
"""cProfile of the host side of predict() on a 512x512 image (200 calls, sync=False)."""
import cProfile, pstats, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_fhd")
kw = dict(frames_per_second=30) if N > 1 else {}
for _ in range(10):
    m.predict(test, ref, sync=False, **kw)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    m.predict(test, ref, sync=False, **kw)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
