#!/usr/bin/env python3
"""How much does K1 depend on where the two clips sit relative to each other?  Both clips are carved out of one allocation,
the reference clip at (size of the test clip + delta) bytes; K1 us/frame per delta."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = 2160, 3840, 30
fps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
t0, r0 = synth_video_pair(N, H, W, device="cuda")
sz = t0.numel()
big = torch.empty(2 * sz + (64 << 20), dtype=torch.uint8, device="cuda")
m = fv.fvvdp(display_name="standard_4k"); m.timing = True
print("big at 0x%x, clip %d bytes (mod 2MiB = %d)" % (big.data_ptr(), sz, sz % (2 << 20)))
for delta in [0, 256, 1024, 4096, 16384, 65536, 1 << 20, (2 << 20) - sz % (2 << 20), (2 << 20) - sz % (2 << 20) + 4096, (2 << 20) - sz % (2 << 20) + 65536,
              (32 << 20) - sz % (32 << 20), 12345 * 256, 7777 * 4096]:
    test = big[:sz].view(t0.shape); ref = big[sz + delta: 2 * sz + delta].view(t0.shape)
    test.copy_(t0); ref.copy_(r0)
    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    m.predict(test, ref, frames_per_second=fps); torch.cuda.synchronize()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    for it in range(3):
        m.predict(test, ref, frames_per_second=fps)
    torch.cuda.synchronize()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    print("delta %10d  (ref-test) mod 2MiB %8d  mod 64KiB %6d: K1 %.1f us/frame" % (delta, (sz + delta) % (2 << 20), (sz + delta) % 65536, ms[0] / (3 * N) * 1e3), flush=True)
