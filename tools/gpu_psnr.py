#!/usr/bin/env python3
"""PU21-PSNR timing probe at 4K."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t8, r8 = synth_video_pair(N, 2160, 3840, device="cuda")
m = fv.pu_psnr(display_name="standard_4k")
for name, (t, r) in {"uint8": (t8, r8), "fp32": (t8[:, :, :20].float() / 255.0, r8[:, :, :20].float() / 255.0)}.items():
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, _ = m.predict(t, r, frames_per_second=30)
        q = float(q); best = min(best, time.perf_counter() - t0)
    n = t.shape[2]
    print("pu_psnr %s 3840x2160x%d: %.2f ms  %.0f Mpix/s  %.4f dB" % (name, n, best * 1e3, 2 * 3840 * 2160 * n / best / 1e6, q), flush=True)
