"""GPU: the two-levels-per-pass pyramid kernel (band2_kernel) against the one-level kernel and the CPU oracle.

By default the library picks band2_kernel only for large levels (>= 0.5 Mpixel); FVVDP_BAND_FUSE=1 forces it wherever
its border logic is valid, =0 disables it, FVVDP_BAND2_KR sets the chunk height (level-C rows), so that strip seams,
chunk seams and all row/column parities of three consecutive levels are exercised at small sizes.  Both kernels
evaluate the same per-pixel expressions; only the order in which the per-wave partial sums are added differs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(H, W, seed, N=None):
    rng = np.random.RandomState(seed)
    shape = (H, W) if N is None else (N, H, W)
    base = rng.randint(0, 256, shape).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    ref = np.clip(0.6 * base + 50 + 40 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 0, 255).astype(np.uint8)
    test = np.clip(ref.astype(np.int32) + rng.randint(-6, 7, shape), 0, 255).astype(np.uint8)
    return test, ref


def _run(monkeypatch, test, ref, fuse, kr=None, disp="standard_4k", **kw):
    import fovvideovdp_amd as fv
    monkeypatch.setenv("FVVDP_BAND_FUSE", str(fuse))
    if kr is None:
        monkeypatch.delenv("FVVDP_BAND2_KR", raising=False)
    else:
        monkeypatch.setenv("FVVDP_BAND2_KR", str(kr))
    q, st = fv.fvvdp(display_name=disp).predict(test, ref, **kw)
    return float(q), st["Q_per_ch"].astype(np.float64)


def _close(a, b, rel=3e-6):
    # measured: <= 6e-7 relative (different summation order of the partial sums only)
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= rel * np.abs(b) + 1e-9 * np.max(np.abs(b))), np.max(np.abs(a - b) / (np.abs(b) + 1e-30))


# widths around the strip pitch (54 level-B = 108 level-A columns) and its multiples, every row/column parity mod 4
IMG_SIZES = [(16, 16), (17, 19), (20, 24), (33, 107), (34, 108), (35, 109), (36, 110), (37, 215), (38, 216), (39, 217),
             (40, 218), (41, 219), (63, 64), (64, 65), (65, 66), (66, 67), (130, 323), (131, 324), (97, 433), (255, 256)]


@pytest.mark.parametrize("H,W", IMG_SIZES)
def test_image_fused_equals_one_level(monkeypatch, H, W):
    test, ref = _pair(H, W, 31 * H + W)
    q0, Q0 = _run(monkeypatch, test, ref, 0, dim_order="HW")
    for kr in (None, 1, 2, 3, 5):
        q1, Q1 = _run(monkeypatch, test, ref, 1, kr, dim_order="HW")
        assert abs(q1 - q0) < 2e-6, (H, W, kr)
        _close(Q1, Q0)


@pytest.mark.parametrize("H,W", [(34, 108), (37, 217), (66, 325), (131, 220), (72, 1000)])
def test_video_fused_equals_one_level_and_oracle(monkeypatch, H, W):
    from oracle import fvvdp_oracle as orc
    N = 5
    test, ref = _pair(H, W, H + 3 * W, N)
    q0, Q0 = _run(monkeypatch, test, ref, 0, disp="standard_fhd", dim_order="FHW", frames_per_second=30)
    for kr in (None, 1, 4):
        q1, Q1 = _run(monkeypatch, test, ref, 1, kr, disp="standard_fhd", dim_order="FHW", frames_per_second=30)
        assert abs(q1 - q0) < 2e-6
        _close(Q1, Q0)
    oq, ost = orc.Oracle("standard_fhd").predict(test, ref, dim_order="FHW", frames_per_second=30)
    assert abs(q1 - float(oq)) < 1e-4
    b = ost["Q_per_ch"].astype(np.float64)
    assert np.all(np.abs(Q1 - b) <= 1e-3 * np.abs(b) + 1e-5 * np.max(b))


def test_large_level_takes_the_two_level_kernel_by_default(monkeypatch):
    """1080p level 0 is above the size threshold: default == forced two-level, bit for bit; and both agree with the
    one-level kernel."""
    from fovvideovdp_amd.synth import synth_video_pair
    test, ref = synth_video_pair(3, 1080, 1920, device="cuda")
    import fovvideovdp_amd as fv
    monkeypatch.delenv("FVVDP_BAND_FUSE", raising=False)
    qd, std = fv.fvvdp(display_name="standard_fhd").predict(test, ref, frames_per_second=30)
    q1, Q1 = _run(monkeypatch, test, ref, 1, disp="standard_fhd", frames_per_second=30)
    q0, Q0 = _run(monkeypatch, test, ref, 0, disp="standard_fhd", frames_per_second=30)
    assert float(qd) == q1 or abs(float(qd) - q1) < 1e-6
    assert abs(q1 - q0) < 2e-6
    _close(Q1, Q0)
    # default takes band2 for level 0+1 only; forced takes it for every pair: the first two bands are bit-identical
    assert np.array_equal(std["Q_per_ch"][:2], Q1[:2].astype(np.float32))


def test_large_still_image_takes_the_two_level_kernel(monkeypatch):
    """A 4K still image (2 planes per pixel, one frame per launch) is above the size threshold as well."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    test, ref = synth_video_pair(1, 2160, 3840, device="cuda")
    t, r = test[0, :, 0].permute(1, 2, 0).contiguous(), ref[0, :, 0].permute(1, 2, 0).contiguous()     # HWC uint8
    monkeypatch.delenv("FVVDP_BAND_FUSE", raising=False)
    qd, sd = fv.fvvdp(display_name="standard_4k").predict(t, r, dim_order="HWC")
    q0, Q0 = _run(monkeypatch, t, r, 0, dim_order="HWC")
    q1, Q1 = _run(monkeypatch, t, r, 1, dim_order="HWC")
    assert abs(float(qd) - q0) < 2e-6 and abs(q1 - q0) < 2e-6
    _close(sd["Q_per_ch"].astype(np.float64), Q0)
    _close(Q1, Q0)
    assert np.array_equal(sd["Q_per_ch"][:2], Q1[:2].astype(np.float32))       # default == forced for levels 0+1


@pytest.mark.parametrize("H,W,N", [(37, 217, None), (66, 325, 5), (40, 1296, None), (72, 1000, 6), (45, 1290, 4), (131, 433, None)])
def test_waves_per_workgroup_change_no_bits(monkeypatch, H, W, N):
    """band2_kernel runs 1, 2, 3 or 4 adjacent strips per workgroup, kept in step by one barrier per stage
    (FVVDP_BAND2_WPB, honoured where it divides the number of strips; default 4 where a level has >= 2560 columns and its strips
    fill whole groups, profiles/r04_lockstep.md).  The work items and the order of their partial sums do not depend on it:
    bit-identical results, also with odd chunk heights.  Sizes: 2, 4, 12, 10, 12 and 5 strips."""
    test, ref = _pair(H, W, 7 * H + W, N)
    kw = dict(dim_order="HW") if N is None else dict(dim_order="FHW", frames_per_second=30)
    for kr in (None, 3):
        monkeypatch.setenv("FVVDP_BAND2_WPB", "1")
        q0, Q0 = _run(monkeypatch, test, ref, 1, kr, **kw)
        for wpb in (2, 3, 4):
            monkeypatch.setenv("FVVDP_BAND2_WPB", str(wpb))
            q1, Q1 = _run(monkeypatch, test, ref, 1, kr, **kw)
            assert q1 == q0 and np.array_equal(Q1, Q0), (H, W, N, kr, wpb)


def test_waves_per_workgroup_default_rule_at_full_size(monkeypatch, capfd):
    """3840x2160 (36 strips at level 1): four waves per workgroup by default; 1920x1080 (18 strips): one.  Same bits as the
    single-wave launch at both sizes."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    monkeypatch.delenv("FVVDP_BAND_FUSE", raising=False)
    monkeypatch.delenv("FVVDP_BAND2_KR", raising=False)
    monkeypatch.setenv("FVVDP_DEBUG_VARIANT", "1")
    for (H, W, disp, want) in ((2160, 3840, "standard_4k", 4), (1080, 1920, "standard_fhd", 1)):
        t, r = synth_video_pair(6, H, W, device="cuda")
        out = []
        for wpb in (None, 1):
            if wpb is None:
                monkeypatch.delenv("FVVDP_BAND2_WPB", raising=False)
            else:
                monkeypatch.setenv("FVVDP_BAND2_WPB", str(wpb))
            capfd.readouterr()
            q, st = fv.fvvdp(display_name=disp).predict(t, r, frames_per_second=30)
            err = capfd.readouterr().err
            if wpb is None:
                assert ("%d waves per workgroup" % want) in err, err[-600:]
            out.append((float(q), st["Q_per_ch"].copy()))
        assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


def test_clamp_free_variant_is_taken_only_where_proven_and_changes_nothing(tmp_path):
    """band2_kernel<P, true> drops the four clamps of the per-pixel tail (L_bkg >= 0.1, contrast <= 1000, the two clamps of the
    CSF table query) when the library has PROVEN from the display model, the RGB->Y weights and the temporal taps that they never
    bind (luminance_range / clamps_never_bind in fvvdp_hip.hip).  Standard-dynamic-range displays qualify; an HDR display with a
    black level below 0.1 cd/m^2 does not, nor does a source that hands over its own luminance frames.  Where it is taken the
    results are bit-identical to FVVDP_BAND_INRANGE=0 (the variable is read once per process)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import fovvideovdp_amd as fv\n"
        "from fovvideovdp_amd.synth import synth_video_pair\n"
        "t, r = synth_video_pair(12, 1080, 1920, device='cuda')\n"
        "out = []\n"
        "def run(tag, m, *a, **k):\n"
        "    sys.stderr.write('CASE %%s\\n' %% tag); sys.stderr.flush()\n"
        "    q, st = m.predict(*a, **k)\n"
        "    out.append(np.concatenate([st['Q_per_ch'].reshape(-1), [float(q)]]))\n"
        "run('sdr_u8', fv.fvvdp(display_name='standard_fhd'), t, r, frames_per_second=30)\n"
        "run('sdr_f32', fv.fvvdp(display_name='standard_fhd'), t.float() / 255, r.float() / 255, frames_per_second=30)\n"
        "run('sdr_bt2020', fv.fvvdp(display_name='standard_fhd', color_space='BT.2020'), t, r, frames_per_second=30)\n"
        "run('sdr_img', fv.fvvdp(display_name='standard_4k'), t[0, :, 0].permute(1, 2, 0), r[0, :, 0].permute(1, 2, 0), dim_order='HWC')\n"
        "run('hdr_pq', fv.fvvdp(display_name='standard_hdr_pq'), t, r, frames_per_second=30)\n"
        "run('sdr_120fps', fv.fvvdp(display_name='standard_fhd'), t, r, frames_per_second=120)\n"
        "mm = fv.fvvdp(display_name='standard_fhd')\n"
        "inner = fv.fvvdp_video_source_array(t, r, 30, display_photometry=mm.display_photometry)\n"
        "class Src(fv.fvvdp_video_source):\n"
        "    def get_video_size(self): return inner.get_video_size()\n"
        "    def get_frames_per_second(self): return 30\n"
        "    def get_test_frame(self, f, device): return inner.get_test_frame(f, device)\n"
        "    def get_reference_frame(self, f, device): return inner.get_reference_frame(f, device)\n"
        "sys.stderr.write('CASE user_src\\n'); sys.stderr.flush()\n"
        "q, st = mm.predict_video_source(Src())\n"
        "out.append(np.concatenate([st['Q_per_ch'].reshape(-1), [float(q)]]))\n"
        "m2 = fv.fvvdp(display_name='standard_fhd', batch_frames=12)\n"       # one context for a user source and then an array
        "sys.stderr.write('CASE user_then\\n'); sys.stderr.flush()\n"
        "m2.predict_video_source(Src())\n"
        "ctx_a = m2._ctx\n"
        "run('array_after_user', m2, t, r, frames_per_second=30)\n"
        "assert m2._ctx is ctx_a\n"
        "np.save(sys.argv[1], np.stack([np.pad(o, (0, 200 - o.size)) for o in out]))\n" % root)
    res, logs = {}, {}
    for mode in ("1", "0"):
        f = str(tmp_path / ("v%s.npy" % mode))
        p = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, FVVDP_BAND_INRANGE=mode, FVVDP_DEBUG_VARIANT="1"),
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        res[mode], logs[mode] = np.load(f), p.stderr
    assert np.array_equal(res["1"], res["0"])                     # dropping clamps that cannot bind changes no bit
    assert "true>" not in logs["0"]

    def variants(log):
        out, tag = {}, None
        for line in log.splitlines():
            if line.startswith("CASE "):
                tag = line.split()[1]
            elif "band2_kernel<" in line and tag:
                out.setdefault(tag, set()).add("true>" in line)
        return out
    v = variants(logs["1"])
    assert v["sdr_u8"] == {True} and v["sdr_f32"] == {True} and v["sdr_bt2020"] == {True} and v["sdr_img"] == {True}
    assert v["sdr_120fps"] == {True}          # 30 taps: the transient plane's range grows with the sum of |taps|, still far from the clamp
    assert v["hdr_pq"] == {False}             # black level 0.0174 cd/m^2 < 0.1: max(L_bkg, 0.1) binds, contrast can reach 1000
    assert v["user_src"] == {False}           # luminance frames from the source's own get_*_frame: range unknown to the library
    assert v["user_then"] == {False} and v["array_after_user"] == {True}     # a call that rewrites every slot starts the range afresh


@pytest.mark.parametrize("H,W,N", [(66, 325, 5), (72, 1000, 6), (131, 433, None), (300, 2600, 3)])
def test_band2_tickets_change_no_bits(monkeypatch, H, W, N):
    """band2_kernel with its work items handed out per XCD at run time (FVVDP_BAND2_TICKET=1: a grid larger than the number of items,
    one atomic ticket per workgroup, stealing between XCDs) against the static split: the same items, partial sums indexed by item ->
    bit-identical, with 1 and 4 waves per workgroup, uniform and two-phase chunking, call after call (the counters are zeroed per launch)."""
    import fovvideovdp_amd as fv
    test, ref = _pair(H, W, 11 * H + W, N)
    kw = dict(dim_order="HW") if N is None else dict(dim_order="FHW", frames_per_second=30)
    monkeypatch.setenv("FVVDP_BAND_FUSE", "1")
    for kr in (None, 2, 5):
        if kr is None:
            monkeypatch.delenv("FVVDP_BAND2_KR", raising=False)
        else:
            monkeypatch.setenv("FVVDP_BAND2_KR", str(kr))
        out = {}
        for t in ("0", "1"):
            monkeypatch.setenv("FVVDP_BAND2_TICKET", t)
            m = fv.fvvdp(display_name="standard_4k")
            res = []
            for rep in range(3):
                q, st = m.predict(test, ref, **kw)
                res.append(np.concatenate([st["Q_per_ch"].reshape(-1), [float(q)]]))
            out[t] = np.stack(res)
        assert np.array_equal(out["0"], out["1"]), (H, W, N, kr)
        assert np.array_equal(out["1"][0], out["1"][2])


@pytest.mark.parametrize("H,W,N", [(67, 119, 4), (66, 118, None), (131, 219, 3), (135, 243, None), (301, 2605, 2), (70, 1002, 3), (73, 333, 5)])
def test_clamp_free_variant_bit_identity_on_odd_sizes_and_ragged_widths(monkeypatch, capfd, H, W, N):
    """band2_kernel<P, true> (no clamps) against <P, false> on odd x odd frames and on widths that are not a multiple of the
    strip pitch (108 level-A columns), forced onto small frames; every halo / invalid lane must stay inside the proven range too:
    bit-identical, and the clamp-free variant really ran."""
    import fovvideovdp_amd as fv
    test, ref = _pair(H, W, 5 * H + W, N)
    kw = dict(dim_order="HW") if N is None else dict(dim_order="FHW", frames_per_second=30)
    monkeypatch.setenv("FVVDP_BAND_FUSE", "1")
    monkeypatch.setenv("FVVDP_DEBUG_VARIANT", "1")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FVVDP_BAND_INRANGE", mode)
        capfd.readouterr()
        q, st = fv.fvvdp(display_name="standard_4k").predict(test, ref, **kw)
        err = capfd.readouterr().err
        out[mode] = (float(q), st["Q_per_ch"].copy(), "true>" in err, "band2_kernel<" in err)
    assert out["1"][3] and out["1"][2] and not out["0"][2], (H, W, N)
    assert out["1"][0] == out["0"][0] and np.array_equal(out["1"][1], out["0"][1]), (H, W, N)


def test_stated_table_range_is_enforced_not_trusted(monkeypatch, capfd):
    """VERDICT r5 #4 / ADVICE r4: a C caller states the range of its code-value table in fvvdp_eotf.L_min / L_max and the pyramid pass
    drops clamps on the strength of it (an unclamped CSF table index among them).  The statement is now ENFORCED where the table
    is read (lut_entry, temporal_kernels.hpp): a wrong or hostile table gives exactly the results of the table clamped to the
    stated range -- finite, identical with and without the clamp-free variant -- and a table that keeps its word is unchanged."""
    import ctypes as C
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.fvvdp import window_frame_indices
    from fovvideovdp_amd.synth import synth_video_pair
    from lowlevel import Pipeline
    H, W, N, fps = 128, 256, 5, 30            # even x even on every level: the parity condition of the proof holds
    test, ref = synth_video_pair(N, H, W, device="cuda")
    m = fv.fvvdp(display_name="standard_fhd")
    m.filter_len = int(np.ceil(250.0 / (1000.0 / fps)))
    F, _ = m.get_temporal_filters(fps)
    taps, fl = F.numpy(), m.filter_len
    idx = window_frame_indices(N, fl, "replicate")
    w = np.asarray([0.2126, 0.7152, 0.0722], dtype=np.float32)
    lut_true = m._code_lut(m.display_photometry, 8).clone()
    lo_t, hi_t = float(lut_true.min()), float(lut_true.max())
    hostile = lut_true.clone()
    hostile[::7] = 0.0
    hostile[3::11] = float("nan")
    hostile[5::13] = -4.0
    hostile[1::17] = 3e7
    monkeypatch.setenv("FVVDP_BAND_FUSE", "1")
    monkeypatch.setenv("FVVDP_DEBUG_VARIANT", "1")

    def run(lut, lo, hi, inrange="1"):
        monkeypatch.setenv("FVVDP_BAND_INRANGE", inrange)
        capfd.readouterr()
        pipe = Pipeline(m, W, H, 4, N)
        e = nat.Eotf()
        e.kind, e.d_lut, e.L_min, e.L_max = nat.EOTF_LUT, lut.data_ptr(), lo, hi
        oob = torch.zeros(1, dtype=torch.int32, device="cuda")
        pipe.temporal(test, ref, nat.FVVDP_U8, 3, N * H * W, H * W, e, w, idx, taps, fl, N, oob=oob)
        Q = pipe.bands_forward(N)
        torch.cuda.synchronize()
        err = capfd.readouterr().err
        pipe.close()
        return Q.cpu().numpy(), ("true>" in err), ("band2_kernel<" in err)

    q_true, var_true, ran2 = run(lut_true, lo_t, hi_t)
    assert ran2 and var_true                                        # a truthful SDR table: the clamp-free variant
    q_unstated, var_u, _ = run(lut_true, 0.0, 0.0)
    assert not var_u and np.array_equal(q_unstated, q_true)         # nothing stated: clamps kept, same bits
    for (lut, lo, hi) in ((lut_true, 5.0, 50.0), (hostile, 1.0, 100.0), (hostile, lo_t, hi_t)):
        q_wrong, var_w, _ = run(lut, lo, hi)
        assert var_w and np.all(np.isfinite(q_wrong))               # the proof goes through on the STATED range ... and holds
        q_clamps, var_c, _ = run(lut, lo, hi, inrange="0")
        assert not var_c and np.array_equal(q_wrong, q_clamps)      # clamp-path-equal
        fixed = torch.clamp(torch.nan_to_num(lut, nan=lo), lo, hi)  # what the statement promised
        q_fixed, _, _ = run(fixed, lo, hi)
        assert np.array_equal(q_wrong, q_fixed)
    q_liar_unstated, var_l, _ = run(hostile, 0.0, 0.0)             # a hostile table without a statement: clamps stay on, entries as they are
    assert not var_l
