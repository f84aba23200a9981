"""Round 4: is a fresh level-0 allocation an independent draw of the temporal kernel's "mode"?  N contexts held at the same time in one
process (each with its own chunk-mapped or hipMalloc scratch), the same 4K x 60 pair through each, K1 / levels 0+1 us per frame from the
library's HIP events.  Round 5: also where each context keeps its level 0 (two chosen ranges by default).
[FVVDP_PLACEMENT_PROBE=0] [FVVDP_ALLOC=malloc] python tools/experiments/gpu_alloc_draws.py [N]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import fovvideovdp_amd as fv  # noqa: E402
from fovvideovdp_amd import _native as nat  # noqa: E402
from fovvideovdp_amd.synth import synth_video_pair  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
t, r = synth_video_pair(60, 2160, 3840, device="cuda")
ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
models = []
for k in range(N):
    m = fv.fvvdp(display_name="standard_4k")
    m.timing = True
    m.predict(t, r, frames_per_second=30)
    models.append(m)
def layout(m):
    st, cm, n, kept = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(-1)
    us = (C.c_float * 8)()
    nat.check(nat.lib().fvvdp_ctx_alloc_info(m._ctx.handle, C.byref(st), C.byref(cm), us, 8, C.byref(n), C.byref(kept)))
    return "%d:%.2f/%.2f" % (cm.value, us[1], us[2])
print("level 0 per context (kind code : TB/s of the pair kept / of the slowest pair; 0 = one range): %s" % "  ".join(layout(m) for m in models))
for rnd in range(2):
    row = []
    for m in models:
        vals = []
        for rep in range(3):
            m.predict(t, r, frames_per_second=30)
            nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
            vals.append((ms[0] / 60 * 1e3, ms[1] / 60 * 1e3))
        row.append("%.2f/%.2f" % (float(np.median([v[0] for v in vals])), float(np.median([v[1] for v in vals]))))
    print("round %d  K1/levels0+1 us per frame per context: %s" % (rnd, "  ".join(row)))
