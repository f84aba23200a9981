import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_image_pair, synth_video_pair
z = np.load('/root/repo/tests/golden/g13_8k.npz')
t, r = synth_image_pair(4320, 7680, 8)
q, st = fv.fvvdp(display_name="standard_4k").predict(t, r, dim_order="HW")
print("img dJOD %.2e rel Q" % (float(q) - float(z["img_jod"])), np.abs(st["Q_per_ch"][:, 0, 0] / z["img_Q_per_ch"][:, 0, 0] - 1))
t, r = synth_video_pair(4, 4320, 7680, device="cuda")
q, st = fv.fvvdp(display_name="standard_4k").predict(t, r, frames_per_second=30)
print("vid dJOD %.2e rel Q max per band" % (float(q) - float(z["vid_jod"])), np.max(np.abs(st["Q_per_ch"] / z["vid_Q_per_ch"] - 1), axis=(1, 2)))
q, st = fv.fvvdp(display_name="standard_4k", foveated=True).predict(t, r, frames_per_second=30)
print("fov dJOD %.2e rel Q max per band" % (float(q) - float(z["fov_jod"])), np.max(np.abs(st["Q_per_ch"] / z["fov_Q_per_ch"] - 1), axis=(1, 2)))
