import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fovvideovdp_amd as fv
from test_gpu_fused import _pair
np.set_printoptions(linewidth=200, precision=3)
def run(fuse, test, ref, gaze, kr=None, **kw):
    os.environ["FVVDP_BAND_FUSE"] = "1"; os.environ["FVVDP_FOV_FUSE"] = str(fuse)
    if kr: os.environ["FVVDP_BAND2_KR"] = str(kr)
    else: os.environ.pop("FVVDP_BAND2_KR", None)
    m = fv.fvvdp(display_name="standard_hdr_pq", foveated=True)
    q, st = m.predict(test, ref, fixation_point=gaze, **kw)
    return float(q), st["Q_per_ch"].astype(np.float64)
for (H, W) in ((64, 120), (65, 121)):
    N = 5
    test, ref = _pair(H, W, 3 * H + W, N)
    for gz in ("move", "centre", "corner"):
        if gz == "move": gaze = np.stack([np.linspace(0, W - 1, N), np.linspace(H - 1, 0, N)], 1).astype(np.float32)
        elif gz == "centre": gaze = np.array([W // 2, H // 2], np.float32)
        else: gaze = np.array([0.0, H - 1.0], np.float32)
        kw = dict(dim_order="FHW", frames_per_second=30)
        q0, Q0 = run(0, test, ref, gaze, **kw)
        q1, Q1 = run(1, test, ref, gaze, **kw)
        rel = np.abs(Q1 - Q0) / (np.abs(Q0) + 1e-9 * np.max(Q0))
        print("NE", os.environ.get("FVVDP_FOV2_NE"), Q0[0, 0], Q1[0, 0])
        print(H, W, gz, "max rel", rel.max(), "bands0-1 rel per frame ch0:", rel[0, 0], rel[1, 0], "other bands max", rel[2:].max())
