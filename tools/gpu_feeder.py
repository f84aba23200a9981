#!/usr/bin/env python3
"""Frame-source probe: the same 1080p x60 clip through (a) the array source, (b) a user source
that returns device-resident luminance frames, (c) a user source that decodes uint8 host frames in its own Python
code (upload + display model per frame)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
H, W, N, fps = (int(os.environ.get("HH", 1080)), int(os.environ.get("WW", 1920)), 60, 30)
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_fhd")
inner = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)
Lt = torch.stack([inner.get_test_frame(f, torch.device("cuda")).reshape(H, W) for f in range(N)])
Lr = torch.stack([inner.get_reference_frame(f, torch.device("cuda")).reshape(H, W) for f in range(N)])
th, rh = test.cpu(), ref.cpu()
inner_h = fv.fvvdp_video_source_array(th, rh, fps, display_photometry=m.display_photometry)

class Resident(fv.fvvdp_video_source):
    def get_video_size(self): return (H, W, N)
    def get_frames_per_second(self): return fps
    def get_test_frame(self, f, device): return Lt[f].view(1, 1, 1, H, W)
    def get_reference_frame(self, f, device): return Lr[f].view(1, 1, 1, H, W)

class Decoding(fv.fvvdp_video_source):
    def get_video_size(self): return (H, W, N)
    def get_frames_per_second(self): return fps
    def get_test_frame(self, f, device): return inner_h.get_test_frame(f, device)
    def get_reference_frame(self, f, device): return inner_h.get_reference_frame(f, device)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); q = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, float(q[0])

a = timeit(lambda: m.predict(test, ref, frames_per_second=fps))
print("array source (device uint8)              : %7.2f ms  JOD %.6f" % a)
for batch in [None if b == "None" else int(b) for b in os.environ.get("BATCHES", "None,4,8,16,60").split(",")]:
    mm = fv.fvvdp(display_name="standard_fhd", batch_frames=batch)
    b = timeit(lambda: mm.predict_video_source(Resident()))
    c = timeit(lambda: mm.predict_video_source(Decoding()))
    print("batch %-4s user source, resident luminance : %7.2f ms (x%.2f)  JOD %.6f | decoding uint8 host frames: %7.2f ms  JOD %.6f" % (
        batch, b[0], b[0] / a[0], b[1], c[0], c[1]))
