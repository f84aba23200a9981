"""Round 6: host time of one queued predict() (sync=False: returns when every launch is enqueued) and its cProfile, 4K x 60 resident pair --
the part of a synchronous step during which the GPU waits for the host."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
t, r = synth_video_pair(60, 2160, 3840, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
for _ in range(5):
    m.predict(t, r, frames_per_second=30)
v = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    q, st = m.predict(t, r, frames_per_second=30, sync=False)
    v.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("host time of predict(sync=False): median %.1f us, min %.1f us" % (np.median(v) * 1e6, np.min(v) * 1e6))
pr = cProfile.Profile()
for _ in range(50):
    torch.cuda.synchronize()
    pr.enable(); q, st = m.predict(t, r, frames_per_second=30, sync=False); pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4000])
