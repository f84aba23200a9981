#!/usr/bin/env python3
"""Where does a user frame source lose time against the array source?  Wall time of predict_video_source next to the sum
of the in-library kernel timings (HIP events), per batch size.  HH/WW/NN/BATCHES env."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair
H, W, N, fps = int(os.environ.get("HH", 2160)), int(os.environ.get("WW", 3840)), int(os.environ.get("NN", 60)), int(os.environ.get("FPS", 30))
test, ref = synth_video_pair(N, H, W, device="cuda")
m0 = fv.fvvdp(display_name="standard_4k")
inner = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m0.display_photometry)
Lt = [inner.get_test_frame(f, torch.device("cuda")).reshape(1, 1, 1, H, W).clone() for f in range(N)]
Lr = [inner.get_reference_frame(f, torch.device("cuda")).reshape(1, 1, 1, H, W).clone() for f in range(N)]
class Resident(fv.fvvdp_video_source):
    def get_video_size(self): return (H, W, N)
    def get_frames_per_second(self): return fps
    def get_test_frame(self, f, device): return Lt[f]
    def get_reference_frame(self, f, device): return Lr[f]
def run(m, fn, label):
    m.timing = True
    fn(); torch.cuda.synchronize()
    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    tot = sum(ms[i] for i in range(18)) / 4
    m.timing = False
    fn(); torch.cuda.synchronize()
    best2 = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best2 = min(best2, time.perf_counter() - t0)
    print("%-28s wall %.2f ms (timing off) %.2f (timing on) | kernels %.2f ms: K1 %.2f (%d launches) bands %.2f" % (
        label, best2 * 1e3, best * 1e3, tot, ms[0] / 4, cnt[0] // 4, sum(ms[i] for i in range(1, 18)) / 4), flush=True)
run(m0, lambda: m0.predict(test, ref, frames_per_second=fps), "array source uint8 RGB")
for b in os.environ.get("BATCHES", "16,30,60").split(","):
    mm = fv.fvvdp(display_name="standard_4k", batch_frames=int(b))
    run(mm, lambda: mm.predict_video_source(Resident()), "user source f32 lum batch %s" % b)
