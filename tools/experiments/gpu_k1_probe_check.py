#!/usr/bin/env python3
"""Online placement selection of level 0 (fvvdp_hip.hip, temporal_channels_core): K1 / K2b per re-created context, call by call
(call 2 times the incumbent buffer, call 3 a fresh one, call 4 keeps the faster), with FVVDP_PLACEMENT_PROBE=1 (default) and =0;
FVVDP_DEBUG_ALLOC=1 prints the comparisons.  Asserts that the results do not depend on the buffer."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = 2160, 3840, 60
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_4k"); m.timing = True
lib = nat.lib()
q0 = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    m._drop_context()
    line = []
    for call in range(8):
        ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
        q, st = m.predict(test, ref, frames_per_second=30); torch.cuda.synchronize()
        nat.check(lib.fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
        line.append("%.1f/%.1f" % (ms[0] / N * 1e3, ms[1] / N * 1e3))
        if q0 is None: q0, Q0 = float(q), st["Q_per_ch"].copy()
        assert float(q) == q0 and np.array_equal(st["Q_per_ch"], Q0), "results must not depend on the placement"
    print("ctx %2d: K1/K2b us/frame per call: %s" % (rep, "  ".join(line)), flush=True)
