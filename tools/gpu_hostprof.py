import os, sys, time, cProfile, pstats, io
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
test, ref = synth_video_pair(60, 2160, 3840, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
for _ in range(3): m.predict(test, ref, frames_per_second=30)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(10): q, st = m.predict(test, ref, frames_per_second=30)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
pr.disable()
print("ms per predict", dt * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
