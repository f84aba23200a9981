#!/usr/bin/env python3
"""K1 (temporal kernel) time against the start of pyramid level 0 INSIDE ONE ALLOCATION (FVVDP_L0_SLACK_MB at context
creation, FVVDP_L0_OFFSET_KB read at every call): separates the virtual / in-allocation offset from the physical placement
of the allocation (which changes with every hipMalloc).  usage: gpu_k1_offset_sweep.py [rounds]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
os.environ["FVVDP_L0_SLACK_MB"] = "1100"
os.environ["FVVDP_DEBUG_ALLOC"] = "1"
import numpy as np, torch
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = 2160, 3840, 60
test, ref = synth_video_pair(N, H, W, device="cuda")
print("src test 0x%x ref 0x%x" % (test.data_ptr(), ref.data_ptr()), flush=True)
m = fv.fvvdp(display_name="standard_4k"); m.timing = True
offs = [int(x) for x in os.environ.get("OFFS", "").split(",") if x] or [0, 4, 64, 256, 1024, 2048, 4096, 8192, 16384, 65536, 129600, 262144, 524288, 1048576]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for r in range(rounds):
    m._drop_context()                      # a new allocation = a new physical placement
    line = []
    for off in offs + [0]:
        os.environ["FVVDP_L0_OFFSET_KB"] = str(off)
        ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
        m.predict(test, ref, frames_per_second=30); torch.cuda.synchronize()
        nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
        for it in range(3): m.predict(test, ref, frames_per_second=30)
        torch.cuda.synchronize()
        nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
        line.append("%d:%.1f/%.1f" % (off, ms[0] / (3 * N) * 1e3, ms[1] / (3 * N) * 1e3))
    print("allocation %d  K1/K2b us/frame by offset KB: %s" % (r, "  ".join(line)), flush=True)
