import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair, synth_gaze
N, H, W = 120, 2160, 3840
z = np.load(os.path.join(ROOT, "tests", "golden", "g4_foveated_uhd_120f.npz"))
test, ref = synth_video_pair(N, H, W, device="cuda")
gaze = synth_gaze(N, H, W)
m = fv.fvvdp(display_name="standard_hdr_pq", foveated=True)
m.timing = True
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q, st = m.predict(test, ref, frames_per_second=30, fixation_point=gaze.numpy())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("config4 4Kx120 foveated PQ: %.1f ms  %.0f Mpix/s  JOD %.6f (golden %.6f, delta %+.2e)" % (dt * 1e3, 2 * W * H * N / dt / 1e6, float(q), float(z["jod"]), float(q) - float(z["jod"])))
qq, gq = st["Q_per_ch"].astype(np.float64), z["Q_per_ch"].astype(np.float64)
print("Q_per_ch worst rel:", np.max(np.abs(qq - gq) / (np.abs(gq) + 1e-6 * gq.max())), "reference CPU seconds:", float(z["seconds"]))
import ctypes as C
from fovvideovdp_amd import _native as nat
ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
nb = m._ctx.key[2]
print("kernel us/frame:", [round(ms[i] / (3 * N) * 1e3, 2) for i in range(nb + 2)], "batch", m._ctx.key[4])
