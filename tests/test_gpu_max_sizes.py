"""GPU: the largest inputs the path is expected to take -- 8K (7680x4320) frames, one pyramid level more than BASELINE's
4K configurations, and clips long enough to run many batches.  Golden g13 holds the REFERENCE's own results for an 8K still
image and a 4-frame 8K video, plain and foveated (tools/gen_golden.py g13; inputs are seeded synthetic data, only outputs are
stored).  The oracle needs ~45 s for ONE 8K frame, so only the still image is also compared with it; beyond that the 8K video and
the long clip are checked through size-independent properties (identical pair -> 10 JOD, vector == scalar temporal kernel,
two-level == one-level pyramid kernel, a clip == the concatenation of its frame ranges)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H8, W8 = 4320, 7680


def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_8k.npz"))


def _check_q(q, gq, fine, coarse):
    """Against the reference's golden at 8K: its lp_norm (torch.norm, fp32 accumulation on the CPU) is itself inexact at this size --
    see tests/test_oracle_golden.py::check_q_large; bound = the usual one + 1.5e-10 per pixel of the band."""
    q, gq = np.asarray(q, np.float64), np.asarray(gq, np.float64)
    assert q.shape == gq.shape
    for b in range(q.shape[0]):
        tol = (fine if b < 3 else coarse) + 1.5e-10 * (H8 * W8) / 4.0 ** b
        assert np.all(np.abs(q[b] - gq[b]) <= tol * np.abs(gq[b]) + 1e-6 * np.max(np.abs(gq))), (b, q[b], gq[b])


def test_8k_image_vs_reference_golden_and_oracle():
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_image_pair
    from oracle import fvvdp_oracle as orc
    z = _golden()
    test, ref = synth_image_pair(H8, W8, 8)
    q, st = fv.fvvdp(display_name="standard_4k").predict(test, ref, dim_order="HW")
    assert abs(float(q) - float(z["img_jod"])) < 3e-5, (float(q), float(z["img_jod"]))           # the reference itself (golden g13); measured 1.1e-5
    _check_q(st["Q_per_ch"][:, 0:1, :], z["img_Q_per_ch"][:, 0:1, :], fine=7e-5, coarse=4e-4)
    oq, ost = orc.Oracle("standard_4k").predict(test, ref, dim_order="HW")
    assert abs(float(q) - float(oq)) < 5e-6, (float(q), float(oq))
    a, b = st["Q_per_ch"][:, 0, 0].astype(np.float64), ost["Q_per_ch"][:, 0, 0].astype(np.float64)
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= 3.5e-4 * np.abs(b) + 1e-5 * np.max(b)), (a, b)


def test_8k_video_vs_reference_golden():
    """4 frames of 8K at 30 fps, plain and foveated, against the reference's results (golden g13: outputs only, the clip is the seeded
    synthetic pair)."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    z = _golden()
    N = int(z["vid_frames"])
    test, ref = synth_video_pair(N, H8, W8, device="cuda")
    q, st = fv.fvvdp(display_name="standard_4k").predict(test, ref, frames_per_second=30)
    assert abs(float(q) - float(z["vid_jod"])) < 5e-5, (float(q), float(z["vid_jod"]))
    _check_q(st["Q_per_ch"], z["vid_Q_per_ch"], fine=7e-5, coarse=4e-4)
    qf, sf = fv.fvvdp(display_name="standard_4k", foveated=True).predict(test, ref, frames_per_second=30)
    assert abs(float(qf) - float(z["fov_jod"])) < 1e-4, (float(qf), float(z["fov_jod"]))
    _check_q(sf["Q_per_ch"], z["fov_Q_per_ch"], fine=1e-3, coarse=2e-3)


def test_8k_video_properties(monkeypatch):
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.synth import synth_video_pair
    N, fps = 6, 30
    test, ref = synth_video_pair(N, H8, W8, device="cuda")
    m = fv.fvvdp(display_name="standard_4k")

    def channels():
        out = torch.empty((N, 4, H8, W8), dtype=torch.float32, device="cuda")
        nat.check(nat.lib().fvvdp_export_level(m._ctx.handle, 0, N, C.c_void_p(out.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    q_same, s_same = m.predict(ref, ref, frames_per_second=fps)
    assert float(q_same) == 10.0 and float(np.max(s_same["Q_per_ch"])) == 0.0

    q, st = m.predict(test, ref, frames_per_second=fps)
    assert 3.0 < float(q) < 10.0 and np.all(np.isfinite(st["Q_per_ch"]))
    r_vec = channels()
    monkeypatch.setenv("FVVDP_TEMPORAL_SCALAR", "1")
    m._drop_context()          # the library reads its switches when a context is created
    q_sca, s_sca = m.predict(test, ref, frames_per_second=fps)
    r_sca = channels()
    monkeypatch.delenv("FVVDP_TEMPORAL_SCALAR")
    m._drop_context()          # the library reads its switches when a context is created
    scale = torch.clamp(r_sca[:, :2].abs(), min=1e-3)
    assert float(((r_vec[:, :2] - r_sca[:, :2]).abs() / scale).max()) < 2e-6
    assert float(((r_vec[:, 2:] - r_sca[:, 2:]).abs() / scale).max()) < 2e-6
    del r_vec, r_sca, scale
    assert abs(float(q) - float(q_sca)) < 1e-5

    monkeypatch.setenv("FVVDP_BAND_FUSE", "0")

    m._drop_context()          # the library reads its switches when a context is created
    q_one, s_one = m.predict(test, ref, frames_per_second=fps)
    monkeypatch.delenv("FVVDP_BAND_FUSE")
    m._drop_context()          # the library reads its switches when a context is created
    assert abs(float(q) - float(q_one)) < 3e-6                              # summation order of the partial sums only
    assert np.allclose(st["Q_per_ch"], s_one["Q_per_ch"], rtol=2e-5, atol=1e-7 * float(np.max(s_one["Q_per_ch"])))

    # foveated evaluation at the same size (the view-direction map and the LUT slices of 8 levels)
    mf = fv.fvvdp(display_name="standard_4k", foveated=True)
    qf_same, _ = mf.predict(ref, ref, frames_per_second=fps)
    qf, sf = mf.predict(test, ref, frames_per_second=fps)
    assert float(qf_same) == 10.0
    assert float(q) - 1e-4 <= float(qf) < 10.0 and np.all(np.isfinite(sf["Q_per_ch"]))   # the periphery hides differences


def test_long_clip_equals_its_frame_ranges():
    """600 frames of 1080p at 60 fps (many batches, source indices far past one batch): the per-frame, per-band sums of the
    whole clip are those of its three frame ranges laid end to end."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    N, H, W, fps = 600, 1080, 1920, 60
    test, ref = synth_video_pair(N, H, W, device="cuda")
    m = fv.fvvdp(display_name="standard_fhd")
    vs = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)
    q, st = m.predict_video_source(vs)
    whole = st["Q_per_ch"]
    assert whole.shape[2] == N and np.all(np.isfinite(whole))
    parts = []
    for lo, hi in ((0, 170), (170, 431), (431, 600)):
        _, sp = m.predict_video_source(vs, frame_range=(lo, hi), pool=False)
        parts.append(sp["Q_per_ch"])
    cat = np.concatenate(parts, axis=2)
    assert cat.shape == whole.shape
    assert np.allclose(cat, whole, rtol=2e-5, atol=1e-7 * float(np.max(whole)))
