#!/usr/bin/env python3
"""cProfile of predict_video_source on a resident user source (1080p x60): where the host time of the feeder goes."""
import cProfile, pstats, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
H, W, N, fps = int(os.environ.get("HH", 1080)), int(os.environ.get("WW", 1920)), 60, 30
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_fhd")
inner = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)
Lt = [inner.get_test_frame(f, torch.device("cuda")).reshape(1, 1, 1, H, W).clone() for f in range(N)]
Lr = [inner.get_reference_frame(f, torch.device("cuda")).reshape(1, 1, 1, H, W).clone() for f in range(N)]
class Resident(fv.fvvdp_video_source):
    def get_video_size(self): return (H, W, N)
    def get_frames_per_second(self): return fps
    def get_test_frame(self, f, device): return Lt[f]
    def get_reference_frame(self, f, device): return Lr[f]
vs = Resident()
for _ in range(3): m.predict_video_source(vs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): m.predict_video_source(vs, sync=False)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host time per call (sync=False, 20 calls): %.3f ms" % ((t1 - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): m.predict_video_source(vs)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
