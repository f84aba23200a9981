import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import fovvideovdp_amd as fv
from test_gpu_fused import _pair
np.set_printoptions(linewidth=200, precision=4)
def run(fuse, test, ref, gaze, **kw):
    os.environ["FVVDP_BAND_FUSE"] = "1"; os.environ["FVVDP_FOV_FUSE"] = str(fuse)
    m = fv.fvvdp(display_name="standard_hdr_pq", foveated=True)
    q, st = m.predict(test, ref, fixation_point=gaze, **kw)
    return float(q), st["Q_per_ch"].astype(np.float64)
H, W, N = 64, 120, 3
test0, ref = _pair(H, W, 3 * H + W, N)
wins = {"all": (0, H, 0, W), "interior": (16, 48, 30, 90), "top rows 0-1": (0, 2, 0, W), "rows 2-5": (2, 6, 0, W), "bottom rows": (H - 2, H, 0, W), "rows H-6..H-2": (H - 6, H - 2, 0, W),
        "left cols 0-1": (0, H, 0, 2), "cols 2-5": (0, H, 2, 6), "right cols": (0, H, W - 2, W), "cols W-6..W-2": (0, H, W - 6, W - 2), "far interior": (40, 60, 80, 110)}
for gname, gaze in (("corner TL", [0.0, 0.0]), ("corner BR", [W - 1.0, H - 1.0]), ("centre", [W / 2, H / 2])):
    for name, (y0, y1, x0, x1) in wins.items():
        test = ref.copy()
        test[:, y0:y1, x0:x1] = test0[:, y0:y1, x0:x1]
        kw = dict(dim_order="FHW", frames_per_second=30)
        q0, Q0 = run(0, test, ref, np.array(gaze, np.float32), **kw)
        q1, Q1 = run(1, test, ref, np.array(gaze, np.float32), **kw)
        rel = np.abs(Q1 - Q0) / (np.abs(Q0) + 1e-30)
        print("%-10s %-16s band0 Q %s rel %s | band1 rel %.1e" % (gname, name, Q0[0, 0, :2], rel[0, 0, :2], rel[1].max()))
