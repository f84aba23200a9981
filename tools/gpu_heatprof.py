#!/usr/bin/env python3
"""cProfile of predict(..., heatmap=...) at 4K: where does the heat-map output path spend its time?"""
import cProfile, pstats, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
kind = sys.argv[2] if len(sys.argv) > 2 else "threshold"
t8, r8 = synth_video_pair(N, 2160, 3840, device="cuda")
m = fv.fvvdp(display_name="standard_4k", heatmap=kind)
m.predict(t8, r8, dim_order="BCFHW", frames_per_second=30)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
q, st = m.predict(t8, r8, dim_order="BCFHW", frames_per_second=30)
torch.cuda.synchronize()
pr.disable()
print("total %.1f ms for %d frames, heatmap %s %s" % ((time.perf_counter() - t0) * 1e3, N, st["heatmap"].shape, st["heatmap"].dtype))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
