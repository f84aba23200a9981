#!/usr/bin/env python3
"""K1 timing robust to buffer placement: the same case repeated with fresh allocations (a random-size spacer tensor shifts the
addresses of the clips), K1 us/frame of each repetition and the median.  A/B: FVVDP_LIB=... per process.
usage: gpu_k1_ab.py fps:N:kind[:reps] ..."""
import ctypes as C, os, sys, random
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W = 2160, 3840
random.seed(1)
for spec in sys.argv[1:]:
    p = spec.split(":"); fps, N, kind = int(p[0]), int(p[1]), p[2]; reps = int(p[3]) if len(p) > 3 else 6
    m = fv.fvvdp(display_name="standard_4k"); m.timing = True
    out = []
    for r in range(reps):
        spacer = torch.empty(random.randrange(1, 64) * 1237 * 4096, dtype=torch.uint8, device="cuda")
        test, ref = synth_video_pair(N, H, W, device="cuda")
        if kind == "u16":
            test = (test.to(torch.int32) * 257).to(torch.int16); ref = (ref.to(torch.int32) * 257).to(torch.int16)
        elif kind == "f32rgb":
            test = test.to(torch.float32) / 255; ref = ref.to(torch.float32) / 255
        elif kind == "f32gray":
            test = test[:, 1:2].to(torch.float32) / 255; ref = ref[:, 1:2].to(torch.float32) / 255
        ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
        m.predict(test, ref, frames_per_second=fps); torch.cuda.synchronize()
        nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))          # drop the warm-up call
        for it in range(3):
            m.predict(test, ref, frames_per_second=fps)
        torch.cuda.synchronize()
        nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
        out.append(ms[0] / (3 * N) * 1e3)
        del test, ref, spacer
    print("%s %d fps %s x%d: K1 median %.1f us/frame  [%s]" % (os.path.basename(nat.LIB_PATH), fps, kind, N, float(np.median(out)),
          " ".join("%.1f" % v for v in out)), flush=True)
