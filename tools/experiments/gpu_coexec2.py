#!/usr/bin/env python3
"""Round 4: where does the co-execution gain of gpu_coexec.py (K1 of one clip against the pyramid pass of another, two
contexts, 4.29 -> 3.79 ms in round 3) go when the two stages share ONE level-0 allocation (two banks) or run on half clips?
Cases: (A) two contexts x 60 frames, (B) one context of 120 slots, K1 -> slots 60..119 against the pass on slots 0..59,
(C) one context of 60 slots in two banks of 30, (D) two contexts x 30 frames.  Each: alone, sum, two streams in both orders."""
import os, sys, time
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
m.filter_len = 8
F, _ = m.get_temporal_filters(fps)
e = nat.Eotf(); lut = m._code_lut(m.display_photometry, 8); e.kind, e.d_lut = nat.EOTF_LUT, lut.data_ptr()
w = [0.2126729, 0.7151522, 0.0721750]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
idx_all = window_frame_indices(N, 8, "replicate")

def timed(fn, reps=6):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts[1:])) * 1e3

def case(name, pk1, slot_k1, pk2, slot_k2, n):
    idx = idx_all[:8 - 1 + n]
    k1 = lambda: pk1.temporal(test, ref, nat.FVVDP_U8, 3, N * H * W, H * W, e, w, idx, F.numpy(), 8, n, slot0=slot_k1)
    k2 = lambda: pk2.bands_forward_at(slot_k2, n)
    # both banks hold valid data
    pk2.temporal(test, ref, nat.FVVDP_U8, 3, N * H * W, H * W, e, w, idx, F.numpy(), 8, n, slot0=slot_k2)
    k1(); k2(); torch.cuda.synchronize()
    a, b = timed(k1), timed(k2)
    def k1_first():
        with torch.cuda.stream(s1): k1()
        with torch.cuda.stream(s2): k2()
    def k2_first():
        with torch.cuda.stream(s2): k2()
        with torch.cuda.stream(s1): k1()
    c, d = timed(k1_first), timed(k2_first)
    print("%-44s K1 %.3f + pass %.3f = %.3f ms | two streams: K1 first %.3f, pass first %.3f  (%.1f %% / %.1f %%)" % (
        name, a, b, a + b, c, d, 100 * (c / (a + b) - 1), 100 * (d / (a + b) - 1)), flush=True)

for rep in range(2):
    pa, pb = Pipeline(m, W, H, 4, N), Pipeline(m, W, H, 4, N)
    case("A two contexts x 60 frames", pa, 0, pb, 0, 60)
    case("D two contexts x 30 frames", pa, 0, pb, 0, 30)
    case("C one context, banks of 30 (slots 30.. | 0..)", pa, 30, pa, 0, 30)
    case("C' same, banks swapped (slots 0.. | 30..)", pa, 0, pa, 30, 30)
    pa.close(); pb.close()
    pc = Pipeline(m, W, H, 4, 2 * N)
    case("B one context of 120 slots (slots 60.. | 0..)", pc, 60, pc, 0, 60)
    pc.close()
