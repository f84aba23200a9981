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
    assert abs(float(q) - float(oq)) < 2e-4, (H, W, float(q), float(oq))
    a, b = st["Q_per_ch"][:, 0, 0].astype(np.float64), ost["Q_per_ch"][:, 0, 0].astype(np.float64)
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= 5e-3 * np.abs(b) + 1e-5 * np.max(b)), (H, W, a, b)


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
    assert abs(float(q) - float(oq)) < 2e-4
    a, b = st["Q_per_ch"].astype(np.float64), ost["Q_per_ch"].astype(np.float64)
    assert np.all(np.abs(a - b) <= 5e-3 * np.abs(b) + 1e-5 * np.max(b)), (H, W)
