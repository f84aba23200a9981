#!/usr/bin/env python3
"""K1 at other frame rates / input types: 4K clips at 30/60/120 fps, uint8 and uint16 RGB (A/B with FVVDP_LIB=...)."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W = 2160, 3840
cases = [(30, 60, "u8"), (60, 120, "u8"), (120, 120, "u8"), (30, 60, "u16")] if len(sys.argv) < 2 else [tuple(a.split(":")) for a in sys.argv[1:]]
for fps, N, kind in cases:
    fps, N = int(fps), int(N)
    test, ref = synth_video_pair(N, H, W, device="cuda")
    if kind == "u16":
        test = (test.to(torch.int32) * 257).to(torch.int16); ref = (ref.to(torch.int32) * 257).to(torch.int16)
    elif kind == "f32gray":
        test = test[:, 1:2].to(torch.float32) / 255; ref = ref[:, 1:2].to(torch.float32) / 255
    elif kind == "f32rgb":
        test = test.to(torch.float32) / 255; ref = ref.to(torch.float32) / 255
    m = fv.fvvdp(display_name=os.environ.get("PROBE_DISPLAY", "standard_4k")); m.timing = True      # PROBE_DISPLAY=standard_hdr_pq: the PQ display model
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, st = m.predict(test, ref, frames_per_second=fps)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    print("%s 4K x%d @%d fps %s: %.2f ms  K1 %.1f us/frame  JOD %.6f" % (os.path.basename(nat.LIB_PATH), N, fps, kind, best * 1e3, ms[0] / (4 * N) * 1e3, float(q)), flush=True)
    del test, ref
