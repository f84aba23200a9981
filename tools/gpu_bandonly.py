#!/usr/bin/env python3
"""Run only stage 2 (fused pyramid levels) a few times on resident 4K x60 data: target for rocprofv3 --pmc passes."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = 2160, 3840, 60
reps = int(os.environ.get("REPS", "3"))
stage = os.environ.get("STAGE", "bands")
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
q, st = m.predict(test, ref, frames_per_second=30)       # fills level 0 of all 60 slots
ctx = m._ctx
Q = torch.zeros((ctx.key[2], 2, N), dtype=torch.float32, device="cuda")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    if stage == "bands":
        nat.check(nat.lib().fvvdp_bands_forward(ctx.handle, N, C.c_void_p(Q.data_ptr()), N, 0, None, None, None, stream))
    else:
        m.predict(test, ref, frames_per_second=30)
torch.cuda.synchronize()
print("stage", stage, "ms per rep", (time.perf_counter() - t0) / reps * 1e3, "Q00", float(Q[0, 0, 0]))
