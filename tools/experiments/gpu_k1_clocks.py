#!/usr/bin/env python3
"""Are the two modes of the temporal kernel (K1 31 vs 36 us/frame, with the pyramid kernel K2b moving the other way) two states
of the power management rather than two memory placements?  For every re-creation of the pyramid scratch: predict() in a loop
for ~1.2 s while a thread samples the clocks and the power the driver reports in sysfs; prints K1 / K2b next to the medians."""
import ctypes as C, glob, os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd import _native as nat
from fovvideovdp_amd.synth import synth_video_pair


def find_sysfs():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(card, "pp_dpm_sclk")):
            continue
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
            if os.path.exists(os.path.join(card, name)):
                out[name] = os.path.join(card, name)
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input"):
                if os.path.exists(os.path.join(hw, name)):
                    out[name] = os.path.join(hw, name)
        break
    return out


def read_one(name, path):
    try:
        txt = open(path).read()
    except OSError:
        return None
    if name.startswith("pp_dpm"):
        for line in txt.splitlines():
            if line.strip().endswith("*"):
                return float("".join(ch for ch in line.split(":")[1] if ch.isdigit() or ch == "."))
        return None
    try:
        return float(txt.strip())
    except ValueError:
        return None


paths = find_sysfs()
print("sysfs:", {k: v for k, v in paths.items()}, flush=True)
H, W, N = 2160, 3840, 60
test, ref = synth_video_pair(N, H, W, device="cuda")
m = fv.fvvdp(display_name="standard_4k"); m.timing = True
lib = nat.lib()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    m._drop_context()
    samples = {k: [] for k in paths}
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            for k, p in paths.items():
                v = read_one(k, p)
                if v is not None:
                    samples[k].append(v)
            time.sleep(0.02)

    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    m.predict(test, ref, frames_per_second=30); torch.cuda.synchronize()
    nat.check(lib.fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 1.2:
        m.predict(test, ref, frames_per_second=30); n += 1
    torch.cuda.synchronize()
    stop.set(); th.join()
    nat.check(lib.fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    med = {k: (np.median(v) if v else float("nan")) for k, v in samples.items()}
    print("ctx %2d: K1 %.1f  K2b %.1f us/frame | %s" % (rep, ms[0] / (n * N) * 1e3, ms[1] / (n * N) * 1e3,
          "  ".join("%s %.0f" % (k.replace("pp_dpm_", "").replace("_input", "").replace("_average", ""), med[k] / (1e6 if k.startswith(("freq", "power")) else 1.0)) for k in sorted(med))), flush=True)
