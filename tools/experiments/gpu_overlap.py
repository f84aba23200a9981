#!/usr/bin/env python3
"""Experiment: does running two half-clips concurrently on two streams (K1 of one against K2 of the other) beat running
the whole clip on one stream?  Two metric objects, two threads, frame_range sharding inside one GPU."""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
N, H, W = 60, 2160, 3840
t, r = synth_video_pair(N, H, W, device="cuda")
m0 = fv.fvvdp(display_name="standard_4k")
vs = fv.fvvdp_video_source_array(t, r, 30, display_photometry=m0.display_photometry)

def whole():
    q, st = m0.predict_video_source(vs)
    return st["Q_per_ch"]

ms = [fv.fvvdp(display_name="standard_4k") for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
out = [None, None]

def half(i, delay):
    with torch.cuda.stream(streams[i]):
        if delay:
            time.sleep(delay)
        q, st = ms[i].predict_video_source(vs, frame_range=(30 * i, 30 * (i + 1)), pool=False)
        out[i] = st["Q_per_ch"]

def split(delay):
    th = [threading.Thread(target=half, args=(i, delay * i)) for i in range(2)]
    for x in th: x.start()
    for x in th: x.join()
    return np.concatenate(out, axis=2)

for name, fn in (("one stream, 60 frames", whole), ("two streams x30, together", lambda: split(0.0)),
                 ("two streams x30, second delayed 1.0 ms", lambda: split(0.001)),
                 ("two streams x30, second delayed 1.5 ms", lambda: split(0.0015))):
    for _ in range(3): fn()
    best = 1e9
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); Q = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("%-42s %.2f ms" % (name, best * 1e3), flush=True)
