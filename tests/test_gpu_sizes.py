"""GPU: sweep of frame sizes around the kernels' internal boundaries (strip width 120/124 fine pixels, odd/even
parities of every pyramid level, tiny frames, very wide and very tall frames) against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(H, W, seed):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (H, W)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    ref = np.clip(0.6 * base + 50 + 40 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 0, 255).astype(np.uint8)
    test = np.clip(ref.astype(np.int32) + rng.randint(-6, 7, (H, W)), 0, 255).astype(np.uint8)
    return test, ref


SIZES = [(8, 8), (9, 13), (16, 17), (31, 33), (5, 200), (200, 6), (17, 123), (18, 124), (19, 125), (33, 243), (34, 244),
         (35, 245), (36, 246), (37, 247), (38, 248), (65, 249), (66, 487), (67, 488), (20, 489), (21, 727), (23, 1000),
         (1000, 23), (129, 257), (255, 255), (256, 256), (257, 255)]


@pytest.mark.parametrize("H,W", SIZES)
def test_image_sizes_vs_oracle(H, W):
    import fovvideovdp_amd as fv
    from oracle import fvvdp_oracle as orc
    test, ref = _pair(H, W, H * 1000 + W)
    m = fv.fvvdp(display_name="standard_4k")
    try:
        oq, ost = orc.Oracle("standard_4k").predict(test, ref, dim_order="HW")
    except Exception:
        oq = None
    if oq is None or ost["Q_per_ch"].shape[0] < 1:
        with pytest.raises(RuntimeError):
            m.predict(test, ref, dim_order="HW")
        return
    q, st = m.predict(test, ref, dim_order="HW")
    assert abs(float(q) - float(oq)) < 5e-6, (H, W, float(q), float(oq))            # measured <= 9.5e-7 over the sweep
    a, b = st["Q_per_ch"][:, 0, 0].astype(np.float64), ost["Q_per_ch"][:, 0, 0].astype(np.float64)
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= 3.5e-4 * np.abs(b) + 1e-5 * np.max(b)), (H, W, a, b)   # measured <= 1.14e-4


@pytest.mark.parametrize("H,W", [(17, 123), (36, 246), (66, 487), (40, 1000)])
def test_video_sizes_vs_oracle(H, W):
    import fovvideovdp_amd as fv
    from oracle import fvvdp_oracle as orc
    N, fps = 4, 30
    frames = [_pair(H, W, 7 * k + H + W) for k in range(N)]
    test = np.stack([f[0] for f in frames], 0)
    ref = np.stack([f[1] for f in frames], 0)
    m = fv.fvvdp(display_name="standard_fhd")
    q, st = m.predict(test, ref, dim_order="FHW", frames_per_second=fps)
    oq, ost = orc.Oracle("standard_fhd").predict(test, ref, dim_order="FHW", frames_per_second=fps)
    assert abs(float(q) - float(oq)) < 5e-6                                         # measured <= 9.5e-7
    a, b = st["Q_per_ch"].astype(np.float64), ost["Q_per_ch"].astype(np.float64)
    assert np.all(np.abs(a - b) <= 3.5e-4 * np.abs(b) + 1e-5 * np.max(b)), (H, W)         # measured <= 1.04e-4


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,bd,css,fps", [(36, 64, 8, "420", 30), (70, 260, 8, "420", 30), (38, 132, 10, "420", 30),
                                            (34, 68, 8, "444", 30), (50, 96, 10, "444", 60), (46, 520, 8, "420", 60),
                                            (8, 8, 8, "420", 30), (10, 12, 10, "420", 30)])
def test_yuv_vector_kernel_equals_scalar_kernel(H, W, bd, css, fps, monkeypatch):
    """The vectorised YUV ingest (4 consecutive pixels per lane, shared chroma taps) and the per-pixel kernel it
    replaces do the same arithmetic in the same order: their pooled band differences must agree to fp32 rounding;
    rows that are not a multiple of the 256-pixel wave tile, odd chroma heights and the image edges are covered.
    (test_yuv_ingest_golden pins the vector kernel against the reference's unpack pipeline; the per-pixel kernel stays
    the path for widths that are not a multiple of 4 and for fl > 16.)"""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_yuv_pair
    N = 7
    ty, ry = synth_yuv_pair(N, H, W, bit_depth=bd, chroma_ss=css)
    m = fv.fvvdp(display_name="standard_fhd")
    vs = fv.fvvdp_video_source_yuv_frames(ty, ry, fps, W, H, bit_depth=bd, chroma_ss=css, color_space="bt709",
                                          display_photometry=m.display_photometry)
    import ctypes as C
    from fovvideovdp_amd import _native as nat

    def channels():           # level 0 of the context = the temporal channels of the batch just processed
        out = torch.empty((N, 4, H, W), dtype=torch.float32, device="cuda")
        nat.check(nat.lib().fvvdp_export_level(m._ctx.handle, 0, N, C.c_void_p(out.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out.cpu().numpy()

    q_vec, s_vec = m.predict_video_source(vs)
    r_vec = channels()
    monkeypatch.setenv("FVVDP_TEMPORAL_SCALAR", "1")
    m._drop_context()          # the library reads its switches when a context is created
    q_sca, s_sca = m.predict_video_source(vs)
    r_sca = channels()
    monkeypatch.delenv("FVVDP_TEMPORAL_SCALAR")
    m._drop_context()          # the library reads its switches when a context is created
    assert np.isfinite(r_vec).all() and r_sca[:, :2].min() > 0
    # per pixel: a few ulp of the sustained luminance (fused-multiply-add contraction may differ between the kernels)
    scale = np.maximum(np.abs(r_sca[:, :2]), 1e-3)
    assert np.max(np.abs(r_vec[:, :2] - r_sca[:, :2]) / scale) < 2e-6
    assert np.max(np.abs(r_vec[:, 2:] - r_sca[:, 2:]) / scale) < 2e-6
    assert np.allclose(s_vec["Q_per_ch"], s_sca["Q_per_ch"], rtol=2e-4, atol=1e-6 * float(np.max(s_sca["Q_per_ch"])))
    assert abs(float(q_vec) - float(q_sca)) < 1e-5


@pytest.mark.parametrize("H,W,bd,css,fps,N", [(40, 72, 8, "420", 120, 34), (36, 66, 10, "420", 25, 9), (30, 50, 8, "444", 50, 14),
                                              (36, 64, 8, "420", 144, 40),       # 36 taps: luminance frames + 64-slot ring (two passes)
                                              (38, 66, 10, "444", 240, 70)])     # 60 taps, 10 bit 4:4:4, width % 4 != 0
def test_yuv_other_frame_rates_and_widths_vs_oracle(H, W, bd, css, fps, N):
    """YUV sources outside the vector kernel's domain -- 120 fps (30 taps: per-pixel kernel with the 32-slot ring),
    widths that are not a multiple of 4 -- and a zero-padded filter (25 fps: 7 taps in the 8-slot ring), against the
    oracle's restatement of the reference's unpack + metric."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_yuv_pair
    from oracle import fvvdp_oracle as orc
    ty, ry = synth_yuv_pair(N, H, W, bit_depth=bd, chroma_ss=css)
    m = fv.fvvdp(display_name="standard_fhd")
    vs = fv.fvvdp_video_source_yuv_frames(ty, ry, fps, W, H, bit_depth=bd, chroma_ss=css, color_space="bt709",
                                          display_photometry=m.display_photometry)
    q, st = m.predict_video_source(vs)
    tn = ty.numpy() if bd == 8 else ty.numpy().astype(np.uint16)
    rn = ry.numpy() if bd == 8 else ry.numpy().astype(np.uint16)
    oq, ost = orc.Oracle("standard_fhd").predict_yuv(tn, rn, fps, W, H, bit_depth=bd, chroma_ss=css, color_space="bt709")
    assert abs(float(q) - float(oq)) < 2e-4, (float(q), float(oq))
    a, b = st["Q_per_ch"].astype(np.float64), ost["Q_per_ch"].astype(np.float64)
    assert np.all(np.abs(a - b) <= 5e-3 * np.abs(b) + 1e-5 * np.max(b))


@pytest.mark.gpu
@pytest.mark.parametrize("fps,kind", [(30, "u8"), (60, "u8"), (120, "u8"), (240, "u8"), (60, "u16"), (120, "f32")])
def test_full_size_temporal_vector_kernel_equals_scalar_kernel(fps, kind, monkeypatch):
    """BASELINE's 4K frame size at the three ring lengths: the register-ring vector kernel (straight-line loop, 4 or 2
    consecutive pixels per lane, LDS transpose) against the per-pixel ring kernel (kept for misaligned sizes) on the same
    clip (240 fps: the 64-slot ring against the generic kernel) -- a size-independent property, the oracle needs minutes at
    this size.  Same arithmetic in the same order:
    the temporal channels agree to a few ulp (the kernels may contract different multiply-adds), pixel by pixel."""
    import ctypes as C
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.synth import synth_video_pair
    N, H, W = 10, 2160, 3840
    test, ref = synth_video_pair(N, H, W, device="cuda")
    if kind == "u16":
        test, ref = (test.to(torch.int32) * 257).to(torch.int16), (ref.to(torch.int32) * 257).to(torch.int16)
    elif kind == "f32":
        test, ref = test.float() / 255, ref.float() / 255
    m = fv.fvvdp(display_name="standard_4k")

    def channels():
        out = torch.empty((N, 4, H, W), dtype=torch.float32, device="cuda")
        nat.check(nat.lib().fvvdp_export_level(m._ctx.handle, 0, N, C.c_void_p(out.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    q_vec, s_vec = m.predict(test, ref, frames_per_second=fps)
    r_vec = channels()
    monkeypatch.setenv("FVVDP_TEMPORAL_SCALAR", "1")
    m._drop_context()          # the library reads its switches when a context is created
    q_sca, s_sca = m.predict(test, ref, frames_per_second=fps)
    r_sca = channels()
    monkeypatch.delenv("FVVDP_TEMPORAL_SCALAR")
    m._drop_context()          # the library reads its switches when a context is created
    assert bool(torch.isfinite(r_vec).all()) and float(r_sca[:, :2].min()) > 0
    scale = torch.clamp(r_sca[:, :2].abs(), min=1e-3)
    assert float(((r_vec[:, :2] - r_sca[:, :2]).abs() / scale).max()) < 2e-6      # sustained channels (test, reference)
    assert float(((r_vec[:, 2:] - r_sca[:, 2:]).abs() / scale).max()) < 2e-6      # transient channels, relative to the luminance
    assert np.allclose(s_vec["Q_per_ch"], s_sca["Q_per_ch"], rtol=1e-4, atol=1e-6 * float(np.max(s_sca["Q_per_ch"])))
    assert abs(float(q_vec) - float(q_sca)) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("disp", ["standard_4k", "standard_hdr_pq", "standard_hdr_linear", "sdr_fhd_24"])
@pytest.mark.parametrize("kind", ["u16", "f32", "f32oob", "u16gray"])
def test_closed_form_display_model_on_pairs_equals_the_per_pixel_kernels(disp, kind, monkeypatch, caplog):
    """Round 6: the register-ring kernels evaluate the closed-form display models (sRGB, gamma, PQ, linear) of 16-bit / float sources on
    (test, reference) PAIRS -- packed instructions, every product and sum rounded like the reference's torch ops
    (fvvdp_display_model.py:147-165, video_source.py:206).  Against the per-pixel kernels (FVVDP_TEMPORAL_SCALAR=1: eotf_one, one stream
    at a time) on the same clip at 30 / 60 / 120 fps (8-, 16- and 32-slot ring): the temporal channels agree to the last bits of a
    luminance (measured <= 2.4e-6 of the luminance, profiles/r06_k1_closed_form.md; PQ at 120 fps is the case an inline-assembly clip
    on a transcendental's result once broke by factors of 1e4), JODs to 1e-5, and samples outside [0,1] raise the reference's warning on
    both paths."""
    import ctypes as C
    import logging
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.synth import synth_video_pair
    N, H, W = 12, 72, 128
    test, ref = synth_video_pair(N, H, W, device="cuda")
    if kind.startswith("u16"):
        test, ref = (test.to(torch.int32) * 257 + 3).clamp(0, 65535).to(torch.int16), (ref.to(torch.int32) * 257).to(torch.int16)
        if kind == "u16gray":
            test, ref = test[:, 1:2].contiguous(), ref[:, 1:2].contiguous()
    else:
        s = 900.0 if "linear" in disp else 1.0
        test, ref = test.float() / 255 * s, ref.float() / 255 * s
        if kind == "f32oob":
            test = test * 1.6 - 0.3            # the synthetic clip spans 0.14 .. 0.89: -0.08 .. 1.12
    for fps in (30, 60, 120):
        m = fv.fvvdp(display_name=disp)

        def run():
            caplog.clear()
            with caplog.at_level(logging.WARNING):
                q, st = m.predict(test, ref, frames_per_second=fps)
            out = torch.empty((N, 4, H, W), dtype=torch.float32, device="cuda")
            nat.check(nat.lib().fvvdp_export_level(m._ctx.handle, 0, N, C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            return float(q), out, any("outside the valid range" in r.message for r in caplog.records)

        q_vec, r_vec, w_vec = run()
        monkeypatch.setenv("FVVDP_TEMPORAL_SCALAR", "1")
        m._drop_context()          # the library reads its switches when a context is created
        q_sca, r_sca, w_sca = run()
        monkeypatch.delenv("FVVDP_TEMPORAL_SCALAR")
        m._drop_context()
        assert bool(torch.isfinite(r_vec).all())
        lum = torch.clamp(r_sca[:, :2].abs(), min=1e-3)
        d = (r_vec - r_sca).abs() / torch.cat([lum, lum], dim=1)
        assert float(d.max()) < 6e-6, (disp, kind, fps, float(d.max()))      # measured <= 2.4e-6
        assert abs(q_vec - q_sca) < 1e-5, (disp, kind, fps)
        # the clipping models (sRGB / gamma / PQ) flag float samples outside [0,1]; linear displays and integer codes never do
        expect = kind == "f32oob" and "linear" not in disp
        assert w_vec == w_sca == expect, (disp, kind, fps, w_vec, w_sca)
