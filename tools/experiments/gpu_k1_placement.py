import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = 2160, 3840, 60
test, ref = synth_video_pair(N, H, W, device="cuda")
print("test ptr 0x%x ref ptr 0x%x" % (test.data_ptr(), ref.data_ptr()))
m = fv.fvvdp(display_name="standard_4k"); m.timing = True
for rep in range(8):
    m._drop_context()
    if rep % 2 == 1:
        junk = torch.empty((rep * 173) << 20, dtype=torch.uint8, device="cuda")   # perturb the allocator state
    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    m.predict(test, ref, frames_per_second=30); torch.cuda.synchronize()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    for it in range(3): m.predict(test, ref, frames_per_second=30)
    torch.cuda.synchronize()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    free, total = torch.cuda.mem_get_info()
    print("ctx %d: K1 %.1f us/frame  K2b %.1f  (free %.1f GB)" % (rep, ms[0] / (3 * N) * 1e3, ms[1] / (3 * N) * 1e3, free / 1e9), flush=True)
