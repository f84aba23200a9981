#!/usr/bin/env python3
"""Does a cheap probe predict the placement mode of K1?  For every re-creation of the pyramid scratch: the temporal kernel on
the real 4K x60 clip (HIP events in the library) next to the same kernel fed from ONE dummy frame (all window indices 0, 16
outputs: the reads stay in cache, the writes go to the scratch) -- if the two correlate, the mode is a property of where the
scratch lies and can be probed at context creation."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = 2160, 3840, 60
test, ref = synth_video_pair(N, H, W, device="cuda")
dummy = torch.randint(0, 255, (2, 3, 1, H, W), dtype=torch.uint8, device="cuda")
m = fv.fvvdp(display_name="standard_4k"); m.timing = True
lib = nat.lib()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
oob = torch.zeros(1, dtype=torch.int32, device="cuda")
junk = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    m._drop_context()
    if rep % 3 == 2:
        junk.append(torch.empty((rep * 97 + 50) << 20, dtype=torch.uint8, device="cuda"))   # perturb the allocator
    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    m.predict(test, ref, frames_per_second=30); torch.cuda.synchronize()
    nat.check(lib.fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    for it in range(3): m.predict(test, ref, frames_per_second=30)
    torch.cuda.synchronize()
    nat.check(lib.fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    k1, k2 = ms[0] / (3 * N) * 1e3, ms[1] / (3 * N) * 1e3
    # probe: one dummy frame per stream, 16 outputs
    ctx = m._ctx
    e = nat.Eotf(); e.kind = nat.EOTF_LUT; e.d_lut = m._code_lut(m.display_photometry, 8).data_ptr()
    w = np.asarray([0.2126, 0.7152, 0.0722], dtype=np.float32)
    n_out, fl = 16, 8
    idx = np.zeros(fl - 1 + n_out, dtype=np.int32)
    taps = np.ones((2, fl), dtype=np.float32) / fl
    pr = []
    for it in range(4):
        nat.check(lib.fvvdp_temporal_channels(ctx.handle, C.c_void_p(dummy[0].data_ptr()), C.c_void_p(dummy[1].data_ptr()), nat.FVVDP_U8, 3,
                                              H * W, H * W, C.byref(e), nat.fptr(w), idx.ctypes.data_as(C.POINTER(C.c_int32)), nat.fptr(taps), fl,
                                              n_out, 0, C.c_void_p(oob.data_ptr()), stream))
        torch.cuda.synchronize()
        nat.check(lib.fvvdp_ctx_timing_read(ctx.handle, ms, cnt, 18, 1))
        pr.append(ms[0] / n_out * 1e3)
    print("ctx %2d: K1 %.1f us/frame  K2b %.1f   probe (dummy frame, 16 outputs) %s us/frame" % (rep, k1, k2, " ".join("%.1f" % p for p in pr[1:])), flush=True)
