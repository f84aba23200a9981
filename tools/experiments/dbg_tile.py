"""Round 5: per-band differences between the tile kernel and the streaming kernels (debugging aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import fovvideovdp_amd as fv
from test_gpu_fused import _pair
from oracle import fvvdp_oracle as orc

for (H, W, N) in [(17, 19, None), (64, 97, None), (63, 64, 4), (130, 323, None)]:
    test, ref = _pair(H, W, 13 * H + W, N)
    kw = dict(dim_order="HW") if N is None else dict(dim_order="FHW", frames_per_second=30)
    res = {}
    for tile in ("0", "100000000"):
        os.environ["FVVDP_BAND_TILE"] = tile
        os.environ["FVVDP_BAND_FUSE"] = "0"
        m = fv.fvvdp(display_name="standard_4k")
        q, st = m.predict(test, ref, **kw)
        res[tile] = (float(q), st["Q_per_ch"].astype(np.float64))
    oq, ost = orc.Oracle("standard_4k").predict(test, ref, **kw)
    o = ost["Q_per_ch"].astype(np.float64)
    a, b = res["0"][1], res["100000000"][1]
    print(H, W, N, "JOD stream %.6f tile %.6f oracle %.6f" % (res["0"][0], res["100000000"][0], float(oq)))
    for band in range(a.shape[0]):
        ra = np.max(np.abs(a[band] - o[band]) / (np.abs(o[band]) + 1e-30))
        rb = np.max(np.abs(b[band] - o[band]) / (np.abs(o[band]) + 1e-30))
        rab = np.max(np.abs(a[band] - b[band]) / (np.abs(a[band]) + 1e-30))
        print("   band %d: stream vs oracle %.2e   tile vs oracle %.2e   tile vs stream %.2e   (Q %.5g)" % (band, ra, rb, rab, a[band].flat[0]))
