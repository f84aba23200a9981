import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair, synth_frame_pair
from oracle import fvvdp_oracle as orc
H, W = 4320, 7680
t, r = synth_video_pair(10, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q, st = m.predict(t, r, frames_per_second=30)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("8K x10 video: %.2f ms %.0f Mpix/s JOD %.6f" % (dt * 1e3, 2 * W * H * 10 / dt / 1e6, float(q)))
m1 = fv.fvvdp(display_name="standard_4k", batch_frames=3)
q1, st1 = m1.predict(t, r, frames_per_second=30)
print("batch 3: JOD %.6f max rel dQ %.2e" % (float(q1), np.max(np.abs(st1["Q_per_ch"] - st["Q_per_ch"]) / np.abs(st["Q_per_ch"]))))
ti, ri = t[:, :, 0:1].cpu().numpy(), r[:, :, 0:1].cpu().numpy()
qi, sti = m.predict(ti, ri, frames_per_second=0)
t0 = time.perf_counter()
oq, ost = orc.Oracle("standard_4k").predict(ti, ri, frames_per_second=0)
print("8K image HIP %.6f oracle %.6f (%.1f s) max rel dQ %.2e" % (float(qi), float(oq), time.perf_counter() - t0,
      np.max(np.abs(sti["Q_per_ch"] - ost["Q_per_ch"]) / np.abs(ost["Q_per_ch"]))))
