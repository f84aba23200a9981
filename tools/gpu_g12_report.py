import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair, synth_gaze
z = np.load("/root/repo/tests/golden/g12_heatmaps_foveated.npz")
N, H, W = 6, 68, 121
test, ref = synth_video_pair(N, H, W); gaze = synth_gaze(N, H, W)
m = fv.fvvdp(display_name="standard_fhd", heatmap="raw", foveated=True)
q, st = m.predict(test, ref, frames_per_second=30, fixation_point=gaze.numpy())
hm, g = st["heatmap"].float().numpy(), z["video_raw"].astype(np.float32)
d = np.abs(hm - g)
print("video: dJOD %.2e  shape %s  max rel %.2e  mean abs %.2e  mean g %.2e" % (abs(float(q) - float(z["video_raw_jod"])), hm.shape, np.max(d / (np.abs(g) + 2e-3)), d.mean(), g.mean()))
t2, r2 = synth_video_pair(1, 135, 240)
m = fv.fvvdp(display_name="standard_hdr_pq", heatmap="raw", foveated=True)
q, st = m.predict(t2[0, :, 0], r2[0, :, 0], dim_order="CHW", fixation_point=np.array([60, 40]))
hm, g = st["heatmap"].float().numpy(), z["image_raw"].astype(np.float32)
d = np.abs(hm - g)
print("image: dJOD %.2e  shape %s  max rel %.2e  mean abs %.2e  mean g %.2e" % (abs(float(q) - float(z["image_raw_jod"])), hm.shape, np.max(d / (np.abs(g) + 2e-3)), d.mean(), g.mean()))
