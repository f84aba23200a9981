#!/usr/bin/env python3
"""Stage-2-only timing (HIP events inside the library) on resident 4K x60 level-0 data; env knobs pass through.
usage: tools/gpu_bandonly_speed.py [reps]"""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = (int(os.environ.get("HH", 2160)), int(os.environ.get("WW", 3840)), int(os.environ.get("NN", 60)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
m.timing = True
q, st = m.predict(test, ref, frames_per_second=30)
ctx = m._ctx
nb = ctx.key[2]
Q = torch.zeros((nb, 2, N), dtype=torch.float32, device="cuda")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
nat.check(nat.lib().fvvdp_ctx_timing_read(ctx.handle, ms, cnt, 18, 1))
per_rep = []
for _ in range(reps):
    nat.check(nat.lib().fvvdp_bands_forward(ctx.handle, N, C.c_void_p(Q.data_ptr()), N, 0, None, None, None, stream))
    torch.cuda.synchronize()
    nat.check(nat.lib().fvvdp_ctx_timing_read(ctx.handle, ms, cnt, 18, 1))
    per_rep.append([ms[i] / N * 1e3 for i in range(1, nb + 2)])
a = np.array(per_rep)
tot = a.sum(axis=1)
print("%s bands us/frame median [%s] | total median %.2f min %.2f max %.2f | Q00 %.6f" % (
    " ".join("%s=%s" % (k, os.environ[k]) for k in ("FVVDP_BAND_FUSE", "FVVDP_BAND2_KR", "FVVDP_BAND_CR", "FVVDP_LIB") if k in os.environ),
    " ".join("%.2f" % x for x in np.median(a, axis=0)), np.median(tot), tot.min(), tot.max(), float(Q[0, 0, 0])), flush=True)
