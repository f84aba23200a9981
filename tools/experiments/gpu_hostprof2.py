#!/usr/bin/env python3
"""Host time of predict(sync=False) per call, by function (tottime, microseconds per call): what stands between two calls."""
import cProfile, pstats, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
H, W, N = int(os.environ.get("HH", 1080)), int(os.environ.get("WW", 1920)), 60
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_fhd")
for _ in range(3): m.predict(test, ref, frames_per_second=30)
torch.cuda.synchronize()
R = 200
t0 = time.perf_counter()
for _ in range(R): m.predict(test, ref, frames_per_second=30, sync=False)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host enqueue per call %.1f us" % ((t1 - t0) / R * 1e6))
t0 = time.perf_counter()
for _ in range(50): m.predict(test, ref, frames_per_second=30)
torch.cuda.synchronize()
print("wall per synchronous call %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(R): m.predict(test, ref, frames_per_second=30, sync=False)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = sorted(((v[2], v[3], v[0], k) for k, v in st.stats.items()), reverse=True)[:32]
for tt, ct, nc, k in rows:
    print("%7.2f us tot %7.2f us cum %4.1f calls  %s:%d %s" % (tt / R * 1e6, ct / R * 1e6, nc / R, os.path.basename(k[0]), k[1], k[2]))
