import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair, synth_gaze
N, H, W = 30, 2160, 3840
test, ref = synth_video_pair(N, H, W, device="cuda")
gaze = synth_gaze(N, H, W)
m = fv.fvvdp(display_name="standard_hdr_pq", foveated=True)
for it in range(3):
    q, st = m.predict(test, ref, frames_per_second=30, fixation_point=gaze.numpy())
print(float(q))
