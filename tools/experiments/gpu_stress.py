#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the oracle (sizes, frame rates, padding, dtypes, colour/gray,
foveated).  Not part of the test suite; prints the worst deviations.  usage: gpu_stress.py [cases] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from oracle import fvvdp_oracle as orc
from fovvideovdp_amd.synth import synth_video_pair
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
# frame size range (default: small frames; HLO/HHI/WLO/WHI in the environment for mid-size sweeps)
HLO, HHI = int(os.environ.get("HLO", 17)), int(os.environ.get("HHI", 150))
WLO, WHI = int(os.environ.get("WLO", 17)), int(os.environ.get("WHI", 260))
NMAX = int(os.environ.get("NMAX", 14))
worst = (0.0, None)
worst_q = (0.0, None)
fails = 0
for case in range(n_cases):
    H, W = int(rng.integers(HLO, HHI)), int(rng.integers(WLO, WHI))
    fps = int(rng.choice([0, 24, 25, 30, 50, 60, 120, 144, 240]))
    N = 1 if fps == 0 else int(rng.integers(2, NMAX))
    pad = str(rng.choice(["replicate", "circular", "pingpong"]))
    dt = str(rng.choice(["u8", "u16", "f32"]))
    C_ch = int(rng.choice([1, 3]))
    disp = str(rng.choice(["standard_4k", "standard_fhd", "standard_hdr_pq", "standard_hmd"]))
    fov = bool(rng.integers(0, 4) == 0)
    t, r = synth_video_pair(N, H, W, pair=int(rng.integers(0, 50)))
    if C_ch == 1:
        t, r = t[:, 1:2], r[:, 1:2]
    tn, rn = t.numpy(), r.numpy()
    if dt == "u16":
        tn, rn = tn.astype(np.uint16) * 257, rn.astype(np.uint16) * 257
    elif dt == "f32":
        tn, rn = tn.astype(np.float32) / np.float32(255), rn.astype(np.float32) / np.float32(255)
    desc = f"{W}x{H}x{N} fps={fps} pad={pad} {dt} C={C_ch} {disp} fov={fov}"
    try:
        m = fv.fvvdp(display_name=disp, temp_padding=pad, foveated=fov)
        fix = np.array([W * 0.3, H * 0.6]) if fov else None
        q, st = m.predict(tn, rn, dim_order="BCFHW", frames_per_second=fps, fixation_point=fix)
        o = orc.Oracle(disp, temp_padding=pad, foveated=fov)
        oq, ost = o.predict(tn, rn, "BCFHW", fps, fix)
    except Exception as e:
        msg = str(e)[:90]
        try:
            orc.Oracle(disp, temp_padding=pad, foveated=fov).predict(tn, rn, "BCFHW", fps, None)
            print("FAIL (HIP raised only)", desc, msg); fails += 1
        except Exception as e2:
            print("both raise:", desc, "|", msg, "|", str(e2)[:60])
        continue
    dq = abs(float(q) - float(oq))
    a, b = st["Q_per_ch"].astype(np.float64), ost["Q_per_ch"].astype(np.float64)
    rel = float(np.max(np.abs(a - b) / (np.abs(b) + 1e-5 * b.max() + 1e-12)))
    if dq > worst[0]: worst = (dq, desc)
    if rel > worst_q[0]: worst_q = (rel, desc)
    if dq > 5e-4 or rel > 2e-2:
        print("FAIL", desc, "dJOD %.2e relQ %.2e" % (dq, rel)); fails += 1
print("cases", n_cases, "fails", fails, "| worst dJOD %.2e (%s) | worst rel Q %.2e (%s)" % (worst[0], worst[1], worst_q[0], worst_q[1]))
