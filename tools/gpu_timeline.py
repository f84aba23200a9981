"""Round 4: where the time inside ONE launch of the dominant kernel goes.  Needs the profiling build
(tools/build_variant.sh timeline "-DBAND2_TIMELINE"; run with FVVDP_LIB=build_variants/timeline.so): every single-wave workgroup
of band2_kernel records its start and end on the 100 MHz wall clock, the XCD and the CU it ran on.  Prints the occupancy over the
launch (resident waves in 40 time bins), the duration of a work item by start time, and the per-XCD spans.
  python tools/gpu_timeline.py [out.npy]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import fovvideovdp_amd as fv  # noqa: E402
from fovvideovdp_amd import _native as nat  # noqa: E402
from fovvideovdp_amd.synth import synth_video_pair  # noqa: E402

H, W, N = int(os.environ.get("HH", 2160)), int(os.environ.get("WW", 3840)), int(os.environ.get("NN", 60))
t, r = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_4k")
for _ in range(3):
    q, st = m.predict(t, r, frames_per_second=30)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["FVVDP_LIB"])
fn = lib.fvvdp_debug_k1_timeline if os.environ.get("KERNEL", "band2") == "k1" else lib.fvvdp_debug_timeline     # KERNEL=k1: the temporal kernel (-DK1_TIMELINE build)
fn.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros((65536, 4), dtype=np.uint64)
assert fn(buf.ctypes.data, 65536) == 0
n = int(np.count_nonzero(buf[:, 1]))
rec = buf[:n]
if len(sys.argv) > 1:
    np.save(sys.argv[1], rec)
t0 = rec[:, 0].astype(np.int64); t1 = rec[:, 1].astype(np.int64)
base = t0.min()
s = (t0 - base) / 100.0; e = (t1 - base) / 100.0          # us
xcc = (rec[:, 2] >> np.uint64(32)).astype(np.int64) & 15
hw = (rec[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
span = e.max()
print("JOD %.6f; %d workgroups, launch span %.1f us (first start to last end), sum of item durations %.0f us -> mean residency %.0f waves" % (
    float(q), n, span, (e - s).sum(), (e - s).sum() / span))
d = e - s
print("item duration us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (d.min(), np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
bins = 40
edges = np.linspace(0, span, bins + 1)
occ = np.zeros(bins)
for k in range(bins):
    lo, hi = edges[k], edges[k + 1]
    occ[k] = (np.clip(np.minimum(e, hi) - np.maximum(s, lo), 0, None)).sum() / (hi - lo)
print("resident waves per %.1f us bin:" % (span / bins))
print(" ".join("%4d" % v for v in occ))
print("items started per bin:")
print(" ".join("%4d" % v for v in np.histogram(s, edges)[0]))
print("median duration of the items STARTED in the bin:")
print(" ".join("%4.0f" % (np.median(d[(s >= edges[k]) & (s < edges[k + 1])]) if np.any((s >= edges[k]) & (s < edges[k + 1])) else 0) for k in range(bins)))
print("per XCD: items, first start, last start, last end, distinct (se, sh, cu)")
for x in sorted(set(xcc.tolist())):
    mk = xcc == x
    print("  xcd %d: %5d items  %7.1f %7.1f %7.1f   %d CUs" % (x, mk.sum(), s[mk].min(), s[mk].max(), e[mk].max(), len(set(zip(se[mk].tolist(), sh[mk].tolist(), cu[mk].tolist())))))
last = np.sort(e)[::-1]
print("time before the end at which only k waves were still running: k=2048: %.1f  1024: %.1f  512: %.1f  256: %.1f  64: %.1f us" % tuple(span - last[k - 1] for k in (2048, 1024, 512, 256, 64)))
first = np.sort(s)
print("time after the start at which k waves had started: k=1024: %.1f 2048: %.1f 3072: %.1f us" % tuple(first[k - 1] for k in (1024, 2048, 3072)))
