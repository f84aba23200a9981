#!/usr/bin/env python3
"""First-contact diagnostics on the GPU box: stage-by-stage deviations HIP vs golden/oracle, printed not asserted."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.lowlevel import Pipeline
from fovvideovdp_amd.synth import synth_video_pair

G = os.path.join(ROOT, "tests", "golden")
print(torch.cuda.get_device_name(0), "cpus", os.cpu_count())
for (H, W, N, fps) in ((135, 240, 10, 30), (68, 121, 12, 60)):
    z = np.load(os.path.join(G, f"g2_video_{H}x{W}_replicate.npz"))
    frames = (0, 1, N - 1)
    m = fv.fvvdp(display_name="standard_fhd")
    pipe = Pipeline(m, W, H, 4, len(frames))
    R = torch.tensor(np.stack([z[f"R_f{ff}"] for ff in frames], 0), device=m.device)
    pipe.load_planar(R)
    Q, maps = pipe.bands_forward(len(frames), want_maps=True)
    torch.cuda.synchronize()
    for lvl in range(pipe.n_bands + 1):
        pass
    base = pipe.export_level(pipe.n_bands, len(frames)).cpu().numpy()
    for fi, ff in enumerate(frames):
        for b in range(pipe.n_bands):
            gb = z[f"band_f{ff}_b{b}"] * (1.0 if b == 0 else 2.0)
            hb = maps[b]["contrast"][fi].cpu().numpy()
            gl, hl = z[f"lbkg_f{ff}_b{b}"], maps[b]["lbkg"][fi].cpu().numpy()
            line = f"{H}x{W} f{ff} b{b}: band abs {np.max(np.abs(hb-gb)):.2e} (scale {np.max(np.abs(gb)):.2e}) lbkg rel {np.max(np.abs(hl-gl)/gl):.2e}"
            for cc in range(2):
                i = cc * pipe.n_bands + b
                gs, hs = z[f"S_f{ff}_i{i}"], maps[b]["S"][fi, cc].cpu().numpy()
                gd, hd = z[f"D_f{ff}_i{i}"].astype(np.float64), maps[b]["D"][fi, cc].cpu().numpy().astype(np.float64)
                line += f" | cc{cc} S rel {np.max(np.abs(hs-gs)/gs):.2e} D px {np.max(np.abs(hd-gd)/(np.abs(gd)+1e-3*gd.max())):.2e} sum {hd.sum()/gd.sum()-1:+.2e}"
            print(line)
        gb = z[f"band_f{ff}_b{pipe.n_bands}"]
        print(f"   base level abs {np.max(np.abs(base[fi]-gb)):.2e} scale {np.max(np.abs(gb)):.2e}")
    q, gq = Q.cpu().numpy(), z["Q_per_ch"][:, :, list(frames)]
    print("Q rel max", np.max(np.abs(q - gq) / (np.abs(gq) + 1e-6 * gq.max())))
    test, ref = synth_video_pair(N, H, W)
    jq, st = m.predict(test, ref, frames_per_second=fps)
    print("end-to-end JOD", float(jq), "golden", float(z["jod"]), "Q rel max", np.max(np.abs(st["Q_per_ch"] - z["Q_per_ch"]) / (np.abs(z["Q_per_ch"]) + 1e-6 * z["Q_per_ch"].max())))

# quick speed probe at 1080p and 4K
for (H, W, N) in ((1080, 1920, 60), (2160, 3840, 60)):
    test, ref = synth_video_pair(N, H, W, device="cuda")
    m = fv.fvvdp(display_name="standard_4k" if H == 2160 else "standard_fhd")
    m.timing = True
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, st = m.predict(test, ref, frames_per_second=30)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{W}x{H}x{N}: {dt*1e3:.2f} ms  {2*W*H*N/dt/1e6:.0f} Mpix/s  JOD {float(q):.5f}")
    import ctypes as C
    from fovvideovdp_amd import _native as nat
    ms = (C.c_float * 18)(); cnt = (C.c_int32 * 18)()
    nat.check(nat.lib().fvvdp_ctx_timing_read(m._ctx.handle, ms, cnt, 18, 1))
    print("  kernel ms (3 runs):", [round(x, 3) for x in ms[:m._ctx.key[2] + 2]], "counts", list(cnt[:m._ctx.key[2] + 2]))
    gz = os.path.join(G, "g3_synth_%s_60f.npz" % ("uhd" if H == 2160 else "fhd"))
    if os.path.isfile(gz):
        z = np.load(gz)
        print("  golden JOD", float(z["jod"]), "delta", float(q) - float(z["jod"]), "Q rel max", np.max(np.abs(st["Q_per_ch"] - z["Q_per_ch"]) / (np.abs(z["Q_per_ch"]) + 1e-6 * z["Q_per_ch"].max())))
    del test, ref, m
    torch.cuda.empty_cache()
