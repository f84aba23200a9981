#!/usr/bin/env python3
"""Measured parity of the HIP path, stage by stage (run on the GPU box; prints markdown -> profiles/rNN_parity.md).
The tolerances in tests/ are set to <= 3x the numbers printed here."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fovvideovdp_amd as fv
from fovvideovdp_amd.synth import synth_video_pair, synth_gaze
from lowlevel import Pipeline
from oracle import fvvdp_oracle as orc
G = os.path.join(ROOT, "tests", "golden")
load = lambda n: np.load(os.path.join(G, n + ".npz"))
rows = []
def row(case, stage, measure, value): rows.append((case, stage, measure, value))
def relq(q, g):
    q, g = np.asarray(q, np.float64), np.asarray(g, np.float64)
    return np.abs(q - g) / (np.abs(g) + 1e-6 * np.max(np.abs(g)))

def stages(z, name, H, W, frames, foveated=False, gaze=None, disp="standard_fhd", exact=None):
    m = fv.fvvdp(display_name=disp, foveated=foveated)
    pipe = Pipeline(m, W, H, 4, len(frames), foveated=foveated)
    R = torch.tensor(np.stack([z[f"R_f{ff}"] for ff in frames], 0), device=m.device)
    pipe.load_planar(R)
    Q, maps = pipe.bands_forward(len(frames), want_maps=True, fixation=None if gaze is None else gaze[list(frames)])
    base = pipe.export_level(pipe.n_bands, len(frames))
    torch.cuda.synchronize()
    nb = pipe.n_bands
    ec, el, es, ed, edm, eds = [], [], [], [], [], []
    ese, esm = [], []
    for fi, ff in enumerate(frames):
        for b in range(nb):
            k = f"band_f{ff}_b{b}"
            if k in z.files:
                gb = z[k] * (1.0 if b == 0 else 2.0)
                ec.append(np.max(np.abs(maps[b]["contrast"][fi].cpu().numpy() - gb)) / max(float(np.max(np.abs(gb))), 1e-3))
            k = f"lbkg_f{ff}_b{b}"
            if k in z.files:
                el.append(np.max(np.abs(maps[b]["lbkg"][fi].cpu().numpy() - z[k]) / z[k]))
            for cc in range(2):
                i = cc * nb + b
                k = f"S_f{ff}_i{i}"
                if k in z.files:
                    r = np.abs(maps[b]["S"][fi, cc].cpu().numpy() - z[k]) / z[k]
                    es.append(np.max(r)); esm.append(np.mean(r))
                    if exact is not None:
                        r2 = np.abs(maps[b]["S"][fi, cc].cpu().numpy() - exact[(ff, b, cc)]) / exact[(ff, b, cc)]
                        ese.append(np.max(r2))
                k = f"D_f{ff}_i{i}"
                if k in z.files:
                    gd, hd = z[k].astype(np.float64), maps[b]["D"][fi, cc].cpu().numpy().astype(np.float64)
                    r = np.abs(hd - gd) / (np.abs(gd) + 1e-3 * np.max(gd) + 1e-12)
                    ed.append(np.max(r)); edm.append(np.mean(r)); eds.append(abs(hd.sum() / gd.sum() - 1))
        k = f"band_f{ff}_b{nb}"
        if k in z.files:
            row(name, "Gaussian base level", "max abs / max", np.max(np.abs(base[fi].cpu().numpy() - z[k])) / np.max(np.abs(z[k])))
    if ec: row(name, "contrast bands", "max abs / band max", max(ec))
    if el: row(name, "L_bkg", "max rel", max(el))
    if es: row(name, "S (CSF) vs reference", "max rel / worst per-map mean", (max(es), float(np.max(esm))))
    if ese: row(name, "S (CSF) vs oracle with fp64 geometry", "max rel", max(ese))
    if ed: row(name, "D per pixel", "max rel (floor 1e-3 max) / worst per-map mean", (max(ed), float(np.max(edm))))
    if eds: row(name, "D band sums", "max rel", max(eds))
    gq = z["Q_per_ch"][:, :, list(frames)]
    r = relq(Q.cpu().numpy(), gq)
    row(name, "Q_per_ch from golden R", "max rel finest 3 bands / all bands", (float(np.max(r[:3])), float(np.max(r))))

for (H, W, N, fps) in ((135, 240, 10, 30), (68, 121, 12, 60)):
    z = load(f"g2_video_{H}x{W}_replicate")
    stages(z, f"g2 video {H}x{W}", H, W, (0, 1, N - 1))
    for pad in ("replicate", "circular", "pingpong"):
        zz = load(f"g2_video_{H}x{W}_{pad}")
        test, ref = synth_video_pair(N, H, W)
        m = fv.fvvdp(display_name="standard_fhd", temp_padding=pad)
        q, st = m.predict(test, ref, frames_per_second=fps)
        r = relq(st["Q_per_ch"], zz["Q_per_ch"])
        row(f"g2 video {H}x{W} {pad}", "end to end", "|dJOD| / Q max rel fine / all", (abs(float(q) - float(zz["jod"])), float(np.max(r[:3])), float(np.max(r))))

# still images
z0 = load("g0_wavy_facade_blur_4k")
from scipy.ndimage import gaussian_filter
ref = z0["ref_u16"]
test = np.stack([gaussian_filter(ref[..., c], 2, mode="nearest", truncate=2.0) for c in range(3)], -1)
q, st = fv.fvvdp(display_name="standard_4k").predict(test, ref, dim_order="HWC")
r = relq(st["Q_per_ch"][:, 0:1], z0["Q_per_ch"][:, 0:1])
row("g0 README image 1024x683", "end to end", "|dJOD| / Q max rel fine / all", (abs(float(q) - float(z0["jod"])), float(np.max(r[:3])), float(np.max(r))))
z1 = load("g1_crop512_blur_fhd")
q, st = fv.fvvdp(display_name="standard_fhd").predict(z1["test_u16"], ref[85:597, 256:768], dim_order="HWC")
r = relq(st["Q_per_ch"][:, 0:1], z1["Q_per_ch"][:, 0:1])
row("g1 crop 512x512", "end to end", "|dJOD| / Q max rel fine / all", (abs(float(q) - float(z1["jod"])), float(np.max(r[:3])), float(np.max(r))))

# full-size synthetic videos
for tag, H, W, disp in (("fhd", 1080, 1920, "standard_fhd"), ("uhd", 2160, 3840, "standard_4k")):
    z = load(f"g3_synth_{tag}_60f")
    test, ref = synth_video_pair(60, H, W, device="cuda")
    q, st = fv.fvvdp(display_name=disp).predict(test, ref, frames_per_second=30)
    r = relq(st["Q_per_ch"], z["Q_per_ch"])
    row(f"g3 synthetic {W}x{H}x60", "end to end", "|dJOD| / Q max rel fine / all", (abs(float(q) - float(z["jod"])), float(np.max(r[:3])), float(np.max(r))))
    del test, ref

# foveated
z = load("g4_foveated_135x240")
N, H, W = 6, 135, 240
gaze = synth_gaze(N, H, W).numpy()
o = orc.Oracle("standard_hdr_pq", foveated=True)
o.geometry.exact_geometry = True
exact = {}
for ff in (0, N - 1):
    o.capture = {}
    nb = orc.band_frequencies(W, H, o.ppd)[0]
    o.process_frame(ff, z[f"R_f{ff}"], nb, orc.band_frequencies(W, H, o.ppd)[1], 2, gaze, (H, W))
    k = 0
    for cc in range(2):
        for b in range(nb):
            exact[(ff, b, cc)] = o.capture["S"][k]; k += 1
stages(z, "g4 foveated 240x135 PQ", H, W, (0, N - 1), foveated=True, gaze=gaze, disp="standard_hdr_pq", exact=exact)
# how far the reference's own S is from the fp64-geometry value (its rounding noise)
noise = []
for (ff, b, cc), e in exact.items():
    noise.append(np.max(np.abs(z[f"S_f{ff}_i{cc * nb + b}"] - e) / e))
row("g4 foveated 240x135 PQ", "reference's S vs the same formula with fp64 geometry", "max rel", max(noise))
test, ref = synth_video_pair(N, H, W)
m = fv.fvvdp(display_name="standard_hdr_pq", foveated=True)
q, st = m.predict(test, ref, frames_per_second=30, fixation_point=gaze)
r = relq(st["Q_per_ch"], z["Q_per_ch"])
row("g4 foveated 240x135 PQ", "end to end vs reference", "|dJOD| / Q max rel", (abs(float(q) - float(z["jod"])), float(np.max(r))))
oe = orc.Oracle("standard_hdr_pq", foveated=True); oe.geometry.exact_geometry = True
oq, ost = oe.predict(test.numpy(), ref.numpy(), frames_per_second=30, fixation_point=gaze)
r = relq(st["Q_per_ch"], ost["Q_per_ch"])
row("g4 foveated 240x135 PQ", "end to end vs oracle with fp64 geometry", "|dJOD| / Q max rel", (abs(float(q) - float(oq)), float(np.max(r))))
z = load("g4_foveated_uhd_120f")
test, ref = synth_video_pair(120, 2160, 3840, device="cuda")
q, st = fv.fvvdp(display_name="standard_hdr_pq", foveated=True).predict(test, ref, frames_per_second=30, fixation_point=synth_gaze(120, 2160, 3840).numpy())
r = relq(st["Q_per_ch"], z["Q_per_ch"])
row("g4 foveated 3840x2160x120 PQ (configs[3])", "end to end vs reference", "|dJOD| / Q max rel", (abs(float(q) - float(z["jod"])), float(np.max(r))))
del test, ref

# size sweep vs the oracle (same sizes as tests/test_gpu_sizes.py)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_sizes as tgs
worst_q, worst_j = 0.0, 0.0
for (H, W) in tgs.SIZES:
    test, ref = tgs._pair(H, W, H * 1000 + W)
    try:
        oq, ost = orc.Oracle("standard_4k").predict(test, ref, dim_order="HW")
    except Exception:
        continue
    if ost["Q_per_ch"].shape[0] < 1:
        continue
    q, st = fv.fvvdp(display_name="standard_4k").predict(test, ref, dim_order="HW")
    a, b = st["Q_per_ch"][:, 0, 0].astype(np.float64), ost["Q_per_ch"][:, 0, 0].astype(np.float64)
    worst_q = max(worst_q, float(np.max(np.abs(a - b) / (np.abs(b) + 1e-5 * np.max(b)))))
    worst_j = max(worst_j, abs(float(q) - float(oq)))
row("image size sweep (26 sizes) vs oracle", "end to end", "|dJOD| max / Q max rel", (worst_j, worst_q))
worst_q, worst_j = 0.0, 0.0
for (H, W) in [(17, 123), (36, 246), (66, 487), (40, 1000)]:
    fr = [tgs._pair(H, W, 7 * k + H + W) for k in range(4)]
    test = np.stack([f[0] for f in fr], 0); ref = np.stack([f[1] for f in fr], 0)
    q, st = fv.fvvdp(display_name="standard_fhd").predict(test, ref, dim_order="FHW", frames_per_second=30)
    oq, ost = orc.Oracle("standard_fhd").predict(test, ref, dim_order="FHW", frames_per_second=30)
    a, b = st["Q_per_ch"].astype(np.float64), ost["Q_per_ch"].astype(np.float64)
    worst_q = max(worst_q, float(np.max(np.abs(a - b) / (np.abs(b) + 1e-5 * np.max(b)))))
    worst_j = max(worst_j, abs(float(q) - float(oq)))
row("video size sweep (4 sizes) vs oracle", "end to end", "|dJOD| max / Q max rel", (worst_j, worst_q))

def fmt(v):
    if isinstance(v, tuple):
        return " / ".join("%.2e" % x for x in v)
    return "%.2e" % v
print("| case | stage | measure | measured |\n|---|---|---|---|")
for r in rows:
    print("| %s | %s | %s | %s |" % (r[0], r[1], r[2], fmt(r[3])))
