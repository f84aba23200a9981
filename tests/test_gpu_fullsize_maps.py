"""GPU: per-band difference maps at BASELINE sizes, and the foveated path at the 4K display geometry.

north_star asks for parity "on the JOD score and per-band difference maps".  The small goldens compare D maps pixel by pixel
(tests/test_gpu_parity.py); at 1920x1080 x60, 3840x2160 x60 and 3840x2160 x120 (foveated) the reference's own run stored, for
EVERY frame, band and temporal channel, the sum, the sum of squares and the maximum of the D map it fed to its pooling
(`dsum`, tools/gen_golden.py Capture._mm = fvvdp.py:454-467).  Here the HIP path writes its D maps (fvvdp_band_maps.d_D through
the C ABI), reduces them on the device and is compared with those figures.  Bounds are <= 3x what was measured on MI355X."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def _record(tag, obj):
    """measured figures next to the asserts: kept under gpurun_out/ when the tests run on the GPU box"""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "fullsize_maps_%s.json" % tag), "w") as f:
            json.dump(obj, f, indent=1)
    except OSError:
        pass
    print(tag, json.dumps(obj))


def d_map_sums(m, test, ref, fps, batch, gaze=None):
    """(sum, sum of squares, max) of every D map of the clip: float64 [N, 2 * bands, 3] in the reference's call order
    (temporal channel major, then band: fvvdp.py:392-393).  Level 0 comes from the temporal kernel on the source clip, the maps
    from fvvdp_bands_forward with only d_D requested; the reductions run on the device in float64."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.fvvdp import window_frame_indices
    from lowlevel import Pipeline
    N, H, W = test.shape[2], test.shape[3], test.shape[4]
    vs = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)
    m.last_h2d_bytes = 0
    feed = m._make_feeder(vs, W, H)
    m.filter_len = int(np.ceil(250.0 / (1000.0 / fps)))
    F, _ = m.get_temporal_filters(fps)
    taps, fl = np.ascontiguousarray(F.numpy(), dtype=np.float32), m.filter_len
    widx = window_frame_indices(N, fl, m.temp_padding)
    pipe = Pipeline(m, W, H, 4, batch, foveated=gaze is not None)
    nb = pipe.n_bands
    oob = torch.zeros(1, dtype=torch.int32, device=m.device)
    out = np.zeros((N, 2 * nb, 3), dtype=np.float64)
    Qall = torch.zeros((nb, 2, N), dtype=torch.float32, device=m.device)
    dmaps = [torch.empty((batch, 2) + pipe.level_size(b)[::-1], dtype=torch.float32, device=m.device) for b in range(nb)]
    maps_arr = (nat.BandMaps * nb)()
    for b in range(nb):
        maps_arr[b].d_D = dmaps[b].data_ptr()
    g = C.byref(m._geom_struct()) if gaze is not None else None
    for b0 in range(0, N, batch):
        n = min(batch, N - b0)
        idx = np.ascontiguousarray(widx[b0:b0 + fl - 1 + n])
        feed(pipe, idx, taps, fl, n, oob, pipe.stream())
        fx = None
        if gaze is not None:
            fxa = np.ascontiguousarray(gaze[b0:b0 + n], dtype=np.float32)
            fx = nat.fptr(fxa)
        nat.check(pipe.lib.fvvdp_bands_forward(pipe.handle, n, C.c_void_p(Qall.data_ptr()), N, b0, fx, g, maps_arr, pipe.stream()))
        for b in range(nb):
            D = dmaps[b][:n].double()
            s1 = D.sum(dim=(2, 3))
            s2 = (D * D).sum(dim=(2, 3))
            mx = dmaps[b][:n].amax(dim=(2, 3)).double()
            r = torch.stack([s1, s2, mx], dim=2).cpu().numpy()          # [n, 2, 3]
            for cc in range(2):
                out[b0:b0 + n, cc * nb + b] = r[:, cc]
    torch.cuda.synchronize()
    pipe.close()
    return out, Qall.cpu().numpy()


def rel_dev(h, g):
    """worst |h - g| / (|g| + 1e-6 * the largest golden value of the same (band, channel) over the clip), per statistic"""
    floor = 1e-6 * np.max(np.abs(g), axis=0, keepdims=True)
    return np.max(np.abs(h - g) / (np.abs(g) + floor), axis=(0, 1))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("tag,H,W,disp,batch", [("fhd", 1080, 1920, "standard_fhd", 20), ("uhd", 2160, 3840, "standard_4k", 10)])
def test_full_size_difference_maps_golden(tag, H, W, disp, batch):
    """BASELINE configs[1] and [2]: every D map of the 60-frame clip against the reference's sum / sum of squares / maximum."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    z = load(f"g3_synth_{tag}_60f")
    test, ref = synth_video_pair(60, H, W, device="cuda")
    m = fv.fvvdp(display_name=disp)
    got, Q = d_map_sums(m, test, ref, 30, batch)
    want = z["dsum"]
    assert got.shape == want.shape
    dev = rel_dev(got, want)
    # the map-writing kernels produce the same pooled values as the product's (band2_kernel) pass
    q, stats = m.predict(test, ref, frames_per_second=30)
    qdev = float(np.max(np.abs(Q - stats["Q_per_ch"]) / (np.abs(stats["Q_per_ch"]) + 1e-6 * stats["Q_per_ch"].max())))
    _record("g3_" + tag, {"sum": dev[0], "sum_sq": dev[1], "max": dev[2], "Q_maps_pass_vs_product_pass": qdev})
    assert dev[0] < DSUM_TOL[tag][0] and dev[1] < DSUM_TOL[tag][1] and dev[2] < DSUM_TOL[tag][2], dev
    assert qdev < 1e-6            # measured 1.2e-7: the map-writing one-level kernels pool the same values as the two-level pass


# worst relative deviation of (sum, sum of squares, max) over all frames, bands and channels.  Measured on MI355X (round 6, session 1;
# profiles/r06_parity.md): fhd 8.5e-5 / 2.4e-4 / 6.1e-4, uhd 9.9e-5 / 2.4e-4 / 1.4e-3 (the maximum is ONE pixel of D ~ contrast^2.4),
# foveated 4.9e-3 / 1.03e-2 / 1.28e-2 (the reference's rho noise, amplified by D ~ S^2.4; Q_per_ch of the same pass 7.1e-4)
DSUM_TOL = {"fhd": (2.5e-4, 7e-4, 1.8e-3), "uhd": (3e-4, 7e-4, 4e-3), "fov": (1.5e-2, 3e-2, 3.8e-2)}


@pytest.mark.timeout(1200)
def test_full_size_difference_maps_foveated_golden():
    """BASELINE configs[3]: 3840x2160 x120, foveated with moving gaze on standard_hdr_pq.  The reference's rho carries the rounding
    noise of its fp32 tangent difference (test_foveated_4k_geometry_corner_golden below), D ~ S^2.4 amplifies it, and on this
    dark HDR content a band's maximum hangs on one pixel: the bounds are wider than in the non-foveated case."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair, synth_gaze
    z = load("g4_foveated_uhd_120f")
    N, H, W = 120, 2160, 3840
    test, ref = synth_video_pair(N, H, W, device="cuda")
    gaze = synth_gaze(N, H, W).numpy()
    m = fv.fvvdp(display_name="standard_hdr_pq", foveated=True)
    got, Q = d_map_sums(m, test, ref, 30, 10, gaze=gaze)
    want = z["dsum"]
    assert got.shape == want.shape
    dev = rel_dev(got, want)
    gq = z["Q_per_ch"].astype(np.float64)
    qdev = float(np.max(np.abs(Q - gq) / (np.abs(gq) + 1e-6 * gq.max())))
    _record("g4_fov", {"sum": dev[0], "sum_sq": dev[1], "max": dev[2], "Q_maps_pass_vs_reference": qdev})
    assert dev[0] < DSUM_TOL["fov"][0] and dev[1] < DSUM_TOL["fov"][1] and dev[2] < DSUM_TOL["fov"][2], dev
    assert qdev < 2e-3


@pytest.mark.timeout(900)
def test_foveated_4k_geometry_corner_golden():
    """The foveated path at the 4K display geometry (75 ppd, delta = 0.0066 deg -- where the reference's finite difference of fp32
    tangents, fvvdp_display_model.py:475-488, cancels hardest), on 270x480 windows cut from the top-left corner of bands 0-2 of a
    3-frame 3840x2160 clip (golden g16: the reference's own L_bkg and S there, gaze inside the window on frame 0 and in the
    opposite corner on frame 2).  Shown, each asserted: (1) the reference's S is as far from its OWN formula evaluated with fp64
    geometry as it is from the kernel; (2) a second fp32 evaluation of the reference's formula (numpy's tan instead of torch's)
    is ALSO that far from it -- the spread is rounding noise of the formula, not a property an implementation could match; (3) the
    kernel (closed form cos d / (cos a cos(a+d)), no cancellation) agrees with the fp64-geometry evaluation an order of magnitude
    better; (4) end to end the kernel's Q_per_ch is within the 2e-3 accepted at 4K x120 of the reference and ~10x closer to the
    fp64-geometry oracle, which itself differs from the reference by as much as the kernel does."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.fvvdp import window_frame_indices
    from fovvideovdp_amd.synth import synth_video_pair
    from lowlevel import Pipeline
    from oracle import fvvdp_oracle as orc
    z = load("g16_foveated_uhd_corner")
    N, H, W, fps = 3, 2160, 3840, 30
    gaze = z["gaze"]
    test, ref = synth_video_pair(N, H, W, device="cuda")
    m = fv.fvvdp(display_name="standard_hdr_pq", foveated=True)
    q, stats = m.predict(test, ref, frames_per_second=fps, fixation_point=gaze)
    qq, gq = stats["Q_per_ch"].astype(np.float64), z["Q_per_ch"].astype(np.float64)
    q_vs_ref = float(np.max(np.abs(qq - gq) / (np.abs(gq) + 1e-6 * gq.max())))
    oe = orc.Oracle("standard_hdr_pq", foveated=True)
    oe.geometry.exact_geometry = True
    oq, ost = oe.predict(test.cpu().numpy(), ref.cpu().numpy(), frames_per_second=fps, fixation_point=gaze)
    eq = ost["Q_per_ch"].astype(np.float64)
    q_vs_exact = float(np.max(np.abs(qq - eq) / (np.abs(eq) + 1e-6 * eq.max())))
    exact_vs_ref = float(np.max(np.abs(eq - gq) / (np.abs(gq) + 1e-6 * gq.max())))

    # S and L_bkg maps of the kernel at full size, through the C ABI
    vs = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)
    feed = m._make_feeder(vs, W, H)
    taps, fl = np.ascontiguousarray(m.F.numpy(), dtype=np.float32), m.filter_len
    widx = window_frame_indices(N, fl, "replicate")
    pipe = Pipeline(m, W, H, 4, N, foveated=True)
    nb = pipe.n_bands
    oob = torch.zeros(1, dtype=torch.int32, device=m.device)
    feed(pipe, np.ascontiguousarray(widx), taps, fl, N, oob, pipe.stream())
    Q = torch.zeros((nb, 2, N), dtype=torch.float32, device=m.device)
    maps_arr = (nat.BandMaps * nb)()
    S, L = [], []
    for b in range(3):
        w, h = pipe.level_size(b)
        S.append(torch.zeros((N, 2, h, w), device=m.device))
        L.append(torch.zeros((N, h, w), device=m.device))
        maps_arr[b].d_S, maps_arr[b].d_lbkg = S[b].data_ptr(), L[b].data_ptr()
    fxa = np.ascontiguousarray(gaze, dtype=np.float32)
    nat.check(pipe.lib.fvvdp_bands_forward(pipe.handle, N, C.c_void_p(Q.data_ptr()), N, 0, nat.fptr(fxa), C.byref(m._geom_struct()),
                                           maps_arr, pipe.stream()))
    torch.cuda.synchronize()
    r0, r1, c0, c1 = [int(v) for v in z["window"]]
    _F = np.float32
    rho_band = orc.band_frequencies(W, H, oe.ppd)[1]

    def window_S(o, b, ff, cc, lbkg):
        """the oracle's CSF query on the window of band b (pyfvvdp/fvvdp.py:424-447 restricted to the window's pixels)"""
        wb, hb = pipe.level_size(b)
        xv = np.linspace(0.5, wb - 0.5, wb).astype(_F)[c0:c1]
        yv = np.linspace(0.5, hb - 0.5, hb).astype(_F)[r0:r1]
        xx, yy = np.meshgrid(xv, yv, indexing="xy")
        vx, vy = o.geometry.pix2view_direction((wb, hb), xx, yy)
        gx, gy = o.geometry.pix2view_direction((W, H), _F(gaze[ff][0]) + _F(0.5), _F(gaze[ff][1]) + _F(0.5))
        ecc = np.sqrt((vx - gx) ** 2 + (vy - gy) ** 2).astype(_F)
        rho = (_F(rho_band[b]) * o.geometry.resolution_magnification(vx, vy)).astype(_F)
        return orc.cached_sensitivity(o.lut[cc], rho, lbkg, ecc)

    o32 = orc.Oracle("standard_hdr_pq", foveated=True)          # the reference's formula in fp32, numpy's libm
    worst = {"hip_vs_ref": [0.0, 0.0], "hip_vs_exact": [0.0, 0.0], "ref_vs_exact": [0.0, 0.0], "numpy32_vs_ref": [0.0, 0.0], "lbkg": 0.0}
    for ff in (0, N - 1):
        for b in range(3):
            gl = z[f"lbkg_f{ff}_b{b}"]
            hl = L[b][ff, r0:r1, c0:c1].cpu().numpy()
            worst["lbkg"] = max(worst["lbkg"], float(np.max(np.abs(hl - gl) / gl)))
            for cc in range(2):
                gs = z[f"S_f{ff}_b{b}_c{cc}"]
                hs = S[b][ff, cc, r0:r1, c0:c1].cpu().numpy()
                es = window_S(oe, b, ff, cc, gl)
                ns = window_S(o32, b, ff, cc, gl)
                for key, a, bb in (("hip_vs_ref", hs, gs), ("hip_vs_exact", hs, es), ("ref_vs_exact", gs, es), ("numpy32_vs_ref", ns, gs)):
                    rel = np.abs(a - bb) / bb
                    worst[key][0] = max(worst[key][0], float(rel.max()))
                    worst[key][1] = max(worst[key][1], float(rel.mean()))
    pipe.close()
    _record("g16_corner", dict(worst, Q_hip_vs_ref=q_vs_ref, Q_hip_vs_exact=q_vs_exact, Q_exact_vs_ref=exact_vs_ref,
                               jod_delta_ref=abs(float(q) - float(z["jod"])), jod_delta_exact=abs(float(q) - float(oq))))
    # measured on MI355X (round 6, session 1): hip_vs_ref max 1.33e-2 / mean 1.90e-3, ref_vs_exact 1.31e-2 / 1.89e-3, hip_vs_exact
    # 8.8e-4 / 3.2e-6, numpy32_vs_ref 2.0e-2 / 1.3e-3, lbkg 3.9e-7; Q_per_ch: hip vs ref 6.66e-4, exact vs ref 6.67e-4, hip vs exact 2.5e-5;
    # JOD: 2.4e-5 from the reference, 1.9e-6 from the fp64-geometry oracle
    assert worst["lbkg"] < 1.2e-6
    # (1) + (2): the reference against its own formula in fp64 (here: max 1.33e-2, mean 1.9e-3) and in fp32 with another libm
    # (max 2.0e-2, mean 1.3e-3) -- measured in the build container against g16, independent of the GPU
    assert worst["ref_vs_exact"][0] > 5e-3 and worst["ref_vs_exact"][1] > 1e-3
    assert worst["numpy32_vs_ref"][0] > 5e-3 and worst["numpy32_vs_ref"][1] > 5e-4
    # the kernel against the reference: the same distance as (1) ...
    assert worst["hip_vs_ref"][0] < 1.5 * worst["ref_vs_exact"][0] and worst["hip_vs_ref"][1] < 1.1 * worst["ref_vs_exact"][1]
    # (3) ... and against the fp64-geometry evaluation an order of magnitude closer on average (the maximum is the pixel under the
    # gaze, where sqrt(ecc) amplifies a 1e-6 deg rounding difference)
    assert worst["hip_vs_exact"][1] < 1e-5 and worst["hip_vs_exact"][1] < 0.01 * worst["ref_vs_exact"][1], worst      # 600x closer on average
    assert worst["hip_vs_exact"][0] < 2.6e-3, worst
    # (4) end to end
    assert abs(float(q) - float(z["jod"])) < 7e-5 and abs(float(q) - float(oq)) < 6e-6
    assert q_vs_ref < 2e-3 and exact_vs_ref > 3e-4          # reference vs its own formula in fp64: 6.7e-4 (build container)
    assert q_vs_exact < 8e-5 and q_vs_exact < 0.12 * exact_vs_ref, (q_vs_exact, exact_vs_ref)      # 26x closer to the fp64 evaluation
