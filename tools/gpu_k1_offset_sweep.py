#!/usr/bin/env python3
"""K1 (temporal kernel) time against the start offset of pyramid level 0 inside its allocation (FVVDP_L0_OFFSET_KB): looks for
the address relation behind the two placement modes (31-32 vs 36-38 us per frame).  Prints one line per offset."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = 2160, 3840, 60
test, ref = synth_video_pair(N, H, W, device="cuda")
print("test ptr 0x%x ref ptr 0x%x" % (test.data_ptr(), ref.data_ptr()))
m = fv.fvvdp(display_name="standard_4k"); m.timing = True
offs = [int(x) for x in sys.argv[1:]] or ([0] + [1 << k for k in range(2, 21)] + [3 << k for k in range(8, 19, 2)])
for off in offs:
    m._drop_context()
    os.environ["FVVDP_L0_OFFSET_KB"] = str(off)
    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    m.predict(test, ref, frames_per_second=30); torch.cuda.synchronize()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    for it in range(3): m.predict(test, ref, frames_per_second=30)
    torch.cuda.synchronize()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    print("offset %9d KB: K1 %.1f us/frame  K2b %.1f" % (off, ms[0] / (3 * N) * 1e3, ms[1] / (3 * N) * 1e3), flush=True)
