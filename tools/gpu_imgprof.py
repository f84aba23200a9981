import cProfile, pstats, os, sys, time, io
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
t, r = synth_video_pair(1, 1080, 1920, device="cuda")
ti, ri = t[0, :, 0].contiguous(), r[0, :, 0].contiguous()
m = fv.fvvdp(display_name="standard_fhd")
for _ in range(5): q, _ = m.predict(ti, ri, dim_order="CHW"); float(q)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): q, _ = m.predict(ti, ri, dim_order="CHW"); float(q)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4200])
