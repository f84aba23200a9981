import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
z = np.load("/root/repo/tests/golden/g9_high_frame_rates.npz")
H, W = 72, 128
wj = wq = 0
for fps, N in ((120, 34), (144, 40), (240, 64)):
    test, ref = synth_video_pair(N, H, W); t1, r1 = synth_video_pair(N, H, W, C=1)
    cases = {"u8": (test, ref, "standard_fhd"), "u16": (test.numpy().astype(np.uint16) * 257, ref.numpy().astype(np.uint16) * 257, "standard_fhd"),
             "f32pq": (test.float() / 255, ref.float() / 255, "standard_hdr_pq"), "f32gray": (t1.float() / 255, r1.float() / 255, "standard_4k")}
    for tag, (t, r, disp) in cases.items():
        m = fv.fvvdp(display_name=disp); q, st = m.predict(t, r, frames_per_second=fps)
        gq = z[f"{tag}_{fps}_Q"].astype(np.float64); qq = st["Q_per_ch"].astype(np.float64)
        dj = abs(float(q) - float(z[f"{tag}_{fps}_jod"])); dq = np.max(np.abs(qq - gq) / (np.abs(gq) + 1e-3 * np.max(gq)))
        wj = max(wj, dj); wq = max(wq, dq); print(fps, tag, "dJOD %.2e dQ %.2e" % (dj, dq))
print("worst dJOD %.2e worst dQ %.2e" % (wj, wq))
