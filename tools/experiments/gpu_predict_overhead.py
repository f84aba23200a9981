"""Round 5: where the host time of a synchronous predict() goes next to the queued call + one copy (bench.py's step path)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair
t, r = synth_video_pair(60, 2160, 3840, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
for _ in range(3):
    m.predict(t, r, frames_per_second=30)


def med(f, n=15):
    v = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); v.append(time.perf_counter() - t0)
    return float(np.median(v)) * 1e3


def a():
    q, st = m.predict(t, r, frames_per_second=30)
def b():
    q, st = m.predict(t, r, frames_per_second=30); float(q)
def c():
    q, st = m.predict(t, r, frames_per_second=30, sync=False); torch.cuda.synchronize()
def d():
    q, st = m.predict(t, r, frames_per_second=30, sync=False); st["result_buffer"].cpu()
def e():
    q, st = m.predict(t, r, frames_per_second=30, sync=False); st["result_buffer"].unsqueeze(0).cpu()[:, -1].tolist()
for name, f in (("predict(sync=True)", a), ("predict(sync=True) + float(q)", b), ("predict(sync=False) + synchronize", c),
                ("predict(sync=False) + result_buffer.cpu()", d), ("bench step (one pair)", e)):
    print("%-45s %.3f ms" % (name, med(f)))


# round 6: which way of waiting for the result costs what (the step path of bench.py against the synchronous predict())
pin = torch.empty(4096, dtype=torch.float32, pin_memory=True)
ev = torch.cuda.Event()


def f():
    q, st = m.predict(t, r, frames_per_second=30, sync=False)
    rb = st["result_buffer"]; pin[:rb.numel()].copy_(rb, non_blocking=True); torch.cuda.current_stream().synchronize()
def g():
    q, st = m.predict(t, r, frames_per_second=30, sync=False)
    rb = st["result_buffer"]; pin[:rb.numel()].copy_(rb, non_blocking=True); ev.record(); ev.synchronize()
def h():
    q, st = m.predict(t, r, frames_per_second=30, sync=False)
    rb = st["result_buffer"]; pin[:rb.numel()].copy_(rb, non_blocking=True); ev.record()
    while not ev.query():
        pass
def i():
    q, st = m.predict(t, r, frames_per_second=30, sync=False); fv.fvvdp.finish(st)
for name, fn in (("sync=False + pinned copy + stream.synchronize()", f), ("sync=False + pinned copy + event.synchronize()", g),
                 ("sync=False + pinned copy + event.query() spin", h), ("sync=False + fvvdp.finish(stats)", i),
                 ("predict(sync=True) again", a), ("bench step (one pair) again", e)):
    print("%-50s %.3f ms" % (name, med(fn)))
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    a()
pr.disable()
sio = io.StringIO()
pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(18)
print(sio.getvalue()[:3500])
