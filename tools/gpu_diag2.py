import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
from oracle import fvvdp_oracle as orc
H, W = 36, 64
for fps, N in ((120, 34), (144, 40), (60, 20)):
    test, ref = synth_video_pair(N, H, W)
    m = fv.fvvdp(display_name="standard_fhd")
    q, st = m.predict(test, ref, frames_per_second=fps)
    o = orc.Oracle("standard_fhd"); o.capture = {}
    oq, ost = o.predict(test.numpy(), ref.numpy(), frames_per_second=fps)
    from fovvideovdp_amd.lowlevel import Pipeline
    import ctypes as C
    from fovvideovdp_amd import _native as nat
    buf = torch.empty((N, 4, H, W), device="cuda")
    nat.check(nat.lib().fvvdp_export_level(m._ctx.handle, 0, N, C.c_void_p(buf.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    R = buf.cpu().numpy()
    Ro = np.stack(o.capture["R"], 0)
    print("fps", fps, "fl", m.filter_len, "JOD", float(q), float(oq), "F diff", np.max(np.abs(m.F.numpy() - o.F)))
    for pl in range(4):
        d = np.abs(R[:, pl] - Ro[:, pl])
        print("   plane", pl, "max abs", d.max(), "scale", np.abs(Ro[:, pl]).max(), "worst frame", int(d.reshape(N, -1).max(1).argmax()))
    qq, gq = st["Q_per_ch"].astype(np.float64), ost["Q_per_ch"].astype(np.float64)
    r = np.abs(qq - gq) / (np.abs(gq) + 1e-6 * gq.max())
    i = np.unravel_index(r.argmax(), r.shape)
    print("   Q worst rel", r.max(), "at", i, qq[i], gq[i])
