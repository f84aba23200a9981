#!/usr/bin/env python3
"""Do the temporal kernel (memory-bound, write-heavy) and the two-level pyramid kernel (VALU-bound) overlap when they
run on two streams?  K1 of one 4K x60 clip against stage 2 of another (two contexts), alone and together."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.fvvdp import window_frame_indices
from fovvideovdp_amd.synth import synth_video_pair
from lowlevel import Pipeline
H, W, N, fps = 2160, 3840, 60, 30
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
pa, pb = Pipeline(m, W, H, 4, N), Pipeline(m, W, H, 4, N)
m.filter_len = 8
F, _ = m.get_temporal_filters(fps)
idx = window_frame_indices(N, 8, "replicate")
e = nat.Eotf(); lut = m._code_lut(m.display_photometry, 8); e.kind, e.d_lut = nat.EOTF_LUT, lut.data_ptr()
w = [0.2126729, 0.7151522, 0.0721750]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def k1(p):
    p.temporal(test, ref, nat.FVVDP_U8, 3, N * H * W, H * W, e, w, idx, F.numpy(), 8, N)
def k2(p):
    return p.bands_forward(N)
for p in (pa, pb):
    k1(p); k2(p)
torch.cuda.synchronize()
def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
a = timed(lambda: k1(pa)); b = timed(lambda: k2(pb))
def both():
    with torch.cuda.stream(s1): k1(pa)
    with torch.cuda.stream(s2): k2(pb)
c = timed(both)
def both_rev():
    with torch.cuda.stream(s2): k2(pb)
    with torch.cuda.stream(s1): k1(pa)
d = timed(both_rev)
print("K1 alone %.2f ms | stage 2 alone %.2f ms | sum %.2f | two streams K1 first %.2f ms, stage 2 first %.2f ms" % (a, b, a + b, c, d))
