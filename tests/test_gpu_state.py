"""GPU tests of state that must never go stale between calls: the native context and the
code-value tables are keyed by value, so changing the display, the parameters or the photometry object between calls
gives the same numbers as a fresh metric object.  The reference re-reads these on every call
(pyfvvdp/fvvdp.py:147-161,209-213,442-447)."""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fv():
    import fovvideovdp_amd
    from fovvideovdp_amd import _native
    assert torch.cuda.is_available()
    _native.lib()
    return fovvideovdp_amd


def _pair(N=6, H=96, W=160, seed=3):
    from fovvideovdp_amd.synth import synth_video_pair
    return synth_video_pair(N, H, W, pair=seed)


def _custom_geometry(fv, ppd_scale):
    # same resolution as standard_4k (=> same frame size and band count for the test clip) but a different distance
    g = fv.fvvdp_display_geometry((3840, 2160), distance_m=0.7472 * ppd_scale, diagonal_size_inches=30)
    return g


def test_set_display_model_a_b_a_equals_fresh_objects(fv):
    test, ref = _pair()
    ga, gb = _custom_geometry(fv, 1.0), _custom_geometry(fv, 0.93)
    from fovvideovdp_amd.fvvdp import band_frequencies
    na, _ = band_frequencies(160, 96, ga.get_ppd())
    nb, _ = band_frequencies(160, 96, gb.get_ppd())
    assert na == nb and ga.get_ppd() != gb.get_ppd()       # the case the old (W,H,n_bands,...) key could not tell apart

    def fresh(g):
        m = fv.fvvdp(display_name="standard_4k", display_geometry=g)
        q, st = m.predict(test, ref, frames_per_second=30)
        return float(q), st["Q_per_ch"].copy(), st["rho_band"].copy()

    fa, fb = fresh(ga), fresh(gb)
    assert fa[0] != fb[0]
    m = fv.fvvdp(display_name="standard_4k", display_geometry=ga)
    for g, want in ((ga, fa), (gb, fb), (ga, fa), (gb, fb)):
        m.set_display_model("standard_4k", display_geometry=g)
        q, st = m.predict(test, ref, frames_per_second=30)
        assert float(q) == want[0]
        assert np.array_equal(st["Q_per_ch"], want[1])
        assert np.array_equal(st["rho_band"], want[2])


def test_named_display_switch_and_parameter_change(fv):
    test, ref = _pair()
    m = fv.fvvdp(display_name="standard_4k")
    q4, _ = m.predict(test, ref, frames_per_second=30)
    m.set_display_model("standard_fhd")
    qf, _ = m.predict(test, ref, frames_per_second=30)
    qf_fresh, _ = fv.fvvdp(display_name="standard_fhd").predict(test, ref, frames_per_second=30)
    assert float(qf) == float(qf_fresh) and float(qf) != float(q4)
    # attribute edits (what a calibration loop does) must reach the kernels as well
    m.mask_c = m.mask_c + 0.1
    q2, _ = m.predict(test, ref, frames_per_second=30)
    m2 = fv.fvvdp(display_name="standard_fhd")
    m2.mask_c = m.mask_c
    q2_fresh, _ = m2.predict(test, ref, frames_per_second=30)
    assert float(q2) == float(q2_fresh) and float(q2) != float(qf)
    # foveated flag flips the CSF table kind inside the context
    m.foveated = True
    q3, _ = m.predict(test, ref, frames_per_second=30)
    m3 = fv.fvvdp(display_name="standard_fhd", foveated=True)
    m3.mask_c = m.mask_c
    q3_fresh, _ = m3.predict(test, ref, frames_per_second=30)
    assert float(q3) == float(q3_fresh)


def test_photometry_objects_never_alias(fv):
    """Sources built from display NAMES create a fresh photometry object each; CPython reuses their id() once they are
    garbage: a table cached by id() served the previous display's luminances."""
    test, ref = _pair(N=1)
    names = ["standard_4k", "standard_hdr_pq", "sdr_4k_30", "iphone_12_pro"]
    want = {}
    for n in names:
        vs = fv.fvvdp_video_source_array(test, ref, 0, display_photometry=n)
        q, _ = fv.fvvdp(display_name="standard_4k").predict_video_source(vs)
        want[n] = float(q)
    assert len(set(want.values())) == len(want)
    m = fv.fvvdp(display_name="standard_4k")
    for rep in range(6):
        for n in want:
            vs = fv.fvvdp_video_source_array(test, ref, 0, display_photometry=n)
            q, _ = m.predict_video_source(vs)
            assert float(q) == want[n], (rep, n)
            del vs
            gc.collect()
    # the PU21-PSNR side metric shares the table cache class
    p = fv.pu_psnr()
    vals = []
    for n in want:
        vs = fv.fvvdp_video_source_array(test, ref, 0, display_photometry=n)
        vals.append(float(p.predict_video_source(vs)[0]))
        del vs
        gc.collect()
    assert len(set(vals)) == len(vals)


def test_user_photometry_edited_in_place(fv):
    class Dim(fv.fvvdp_display_photometry):
        def __init__(self, peak):
            self.peak = peak

        def forward(self, V):
            return self.peak * V.clamp(0, 1) ** 2.2 + 0.1

        def get_peak_luminance(self):
            return self.peak

        def get_black_level(self):
            return 0.1

    test, ref = _pair(N=1)
    ph = Dim(100.0)
    m = fv.fvvdp(display_name="standard_4k", display_photometry=ph)
    q100, _ = m.predict(test, ref)
    ph.peak = 400.0
    q400, _ = m.predict(test, ref)
    q400_fresh, _ = fv.fvvdp(display_name="standard_4k", display_photometry=Dim(400.0)).predict(test, ref)
    assert float(q400) == float(q400_fresh) and float(q400) != float(q100)


def test_mixed_dtypes_follow_each_arrays_own_dtype(fv):
    """float32 test against uint8 reference: the reference converts each array by its own dtype
    (video_source.py:186-200); the kernel must never read one buffer with the other's element size."""
    test, ref = _pair()
    m = fv.fvvdp(display_name="standard_4k")
    q_u8, st_u8 = m.predict(test, ref, frames_per_second=30)
    q_mix, st_mix = m.predict(test.to(torch.float32) / 255, ref, frames_per_second=30)
    q_mix2, _ = m.predict(test, ref.to(torch.float32) / 255, frames_per_second=30)
    assert abs(float(q_mix) - float(q_u8)) < 1e-4 and abs(float(q_mix2) - float(q_u8)) < 1e-4
    assert np.allclose(st_mix["Q_per_ch"], st_u8["Q_per_ch"], rtol=2e-4, atol=1e-5 * float(st_u8["Q_per_ch"].max()))
    p = fv.pu_psnr()
    a = float(p.predict(test, ref, frames_per_second=30)[0])
    b = float(p.predict(test.to(torch.float32) / 255, ref, frames_per_second=30)[0])
    assert abs(a - b) < 2e-3


@pytest.mark.parametrize("H,W", [(70, 120), (37, 53)])       # 37x53: frames 4 bytes apart modulo 16 -> scalar temporal kernel
@pytest.mark.parametrize("pad,fps,N", [("replicate", 30, 21), ("circular", 30, 13), ("pingpong", 60, 19), ("replicate", 60, 30)])
def test_pipelined_source_feeder_fetches_every_frame_once_and_matches_the_array_path(fv, pad, fps, N, H, W):
    """User video sources go through their own get_*_frame: side-stream, double-buffered, no host
    synchronisation per batch; with `replicate` padding every frame is fetched exactly once, like the reference does
    (pyfvvdp/fvvdp.py:287-288).  Results equal the array path up to the float-vs-table display model (1e-6)."""
    from fovvideovdp_amd.synth import synth_video_pair
    from oracle import fvvdp_oracle as orc
    test, ref = synth_video_pair(N, H, W)
    m = fv.fvvdp(display_name="standard_fhd", temp_padding=pad)
    q0, s0 = m.predict(test, ref, frames_per_second=fps)
    # the parity statement of this row: the CPU oracle on the same clip, at the array path's tolerances
    oq, ost = orc.Oracle("standard_fhd", temp_padding=pad).predict(test.numpy(), ref.numpy(), frames_per_second=fps)
    inner = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)
    calls = {"t": [], "r": []}

    class Src(fv.fvvdp_video_source):
        def get_video_size(self):
            return inner.get_video_size()

        def get_frames_per_second(self):
            return fps

        def get_test_frame(self, frame, device):
            calls["t"].append(frame)
            return inner.get_test_frame(frame, device)

        def get_reference_frame(self, frame, device):
            calls["r"].append(frame)
            return inner.get_reference_frame(frame, device)

    for batch in (None, 3, 5, N):
        calls["t"].clear(); calls["r"].clear()
        mm = fv.fvvdp(display_name="standard_fhd", temp_padding=pad, batch_frames=batch)
        q1, s1 = mm.predict_video_source(Src())
        assert abs(float(q1) - float(oq)) < 2e-5, (pad, batch, float(q1), float(oq))      # vs the oracle (array path: <= 7e-6 measured)
        o = ost["Q_per_ch"].astype(np.float64)
        assert np.all(np.abs(s1["Q_per_ch"] - o) <= 1e-3 * np.abs(o) + 1e-6 * np.max(o)), (pad, batch)   # check_q's end-to-end bound
        assert abs(float(q1) - float(q0)) < 5e-6, (pad, batch)
        a, b = s1["Q_per_ch"].astype(np.float64), s0["Q_per_ch"].astype(np.float64)
        assert np.all(np.abs(a - b) <= 2e-4 * np.abs(b) + 1e-6 * np.max(b)), (pad, batch)
        assert sorted(calls["t"]) == sorted(calls["r"])
        if pad == "replicate":
            assert sorted(calls["t"]) == list(range(N)), (batch, len(calls["t"]))   # each frame exactly once
        else:
            # the padded history of the first window revisits frames that are fetched again later (circular also
            # never shows frame 0, pyfvvdp/fvvdp.py:258-291): bounded by one extra window per batch boundary
            assert set(calls["t"]) <= set(range(N)) and len(calls["t"]) <= 2 * N + int(np.ceil(0.25 * fps))
    # without a host synchronisation per call as well
    q2, s2 = m.predict_video_source(Src(), sync=False)
    fv.fvvdp.finish(s2)
    assert abs(float(q2) - float(q0)) < 5e-6


@pytest.mark.parametrize("fps", [144, 300])
def test_user_source_with_a_filter_longer_than_32_taps_and_host_frames(fv, fps):
    """144 fps -> 36 taps: the 64-slot ring behind fvvdp_temporal_channels_frames; 300 fps -> 75 taps: beyond it (the feeder
    stacks the frames and takes the general entry point, generic kernel); and a source that returns host float64 tensors
    of another shape (converted on the way in)."""
    from fovvideovdp_amd.synth import synth_video_pair
    N, H, W = 40, 36, 64
    test, ref = synth_video_pair(N, H, W)
    m = fv.fvvdp(display_name="standard_fhd")
    q0, s0 = m.predict(test, ref, frames_per_second=fps)
    inner = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)

    class Src(fv.fvvdp_video_source):
        def get_video_size(self):
            return (H, W, N)

        def get_frames_per_second(self):
            return fps

        def get_test_frame(self, frame, device):
            return inner.get_test_frame(frame, torch.device("cpu")).to(torch.float64).reshape(H, W)

        def get_reference_frame(self, frame, device):
            return inner.get_reference_frame(frame, torch.device("cpu")).to(torch.float64).reshape(H, W)

    q1, s1 = m.predict_video_source(Src())
    from oracle import fvvdp_oracle as orc
    oq, ost = orc.Oracle("standard_fhd").predict(test.numpy(), ref.numpy(), frames_per_second=fps)
    assert abs(float(q1) - float(oq)) < 1e-4, (float(q1), float(oq))          # the bound of test_other_frame_rates_vs_oracle
    o = ost["Q_per_ch"].astype(np.float64)
    assert np.all(np.abs(s1["Q_per_ch"] - o) <= 4e-3 * np.abs(o) + 1e-6 * np.max(o))
    assert abs(float(q1) - float(q0)) < 5e-6
    a, b = s1["Q_per_ch"].astype(np.float64), s0["Q_per_ch"].astype(np.float64)
    assert np.all(np.abs(a - b) <= 4e-3 * np.abs(b) + 1e-6 * np.max(b))      # 36x64: every band is tiny


def test_source_that_reuses_its_output_buffer(fv):
    """a source that decodes every frame into ONE buffer per stream (`return self._buf`).  The zero-copy feeder holds
    the returned tensors by reference for the whole temporal window; it must notice the recycled address, copy from then on
    and re-fetch what the reuse overwrote -- the result equals the oracle's / the array path's, not a window of identical
    frames.  Also a ring of 3 buffers (shorter than the window) and the same clip with sync=False."""
    from fovvideovdp_amd.synth import synth_video_pair
    from oracle import fvvdp_oracle as orc
    N, H, W, fps = 14, 70, 120, 30
    test, ref = synth_video_pair(N, H, W)
    m = fv.fvvdp(display_name="standard_fhd")
    q0, s0 = m.predict(test, ref, frames_per_second=fps)
    oq, ost = orc.Oracle("standard_fhd").predict(test.numpy(), ref.numpy(), frames_per_second=fps)
    inner = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)

    class Reuser(fv.fvvdp_video_source):
        def __init__(self, k):
            self.k = k
            self.bt = [torch.empty((1, 1, 1, H, W), device="cuda") for _ in range(k)]
            self.br = [torch.empty((1, 1, 1, H, W), device="cuda") for _ in range(k)]
            self.n = 0

        def get_video_size(self):
            return (H, W, N)

        def get_frames_per_second(self):
            return fps

        def get_test_frame(self, frame, device):
            b = self.bt[frame % self.k]
            b.copy_(inner.get_test_frame(frame, device))
            self.n += 1
            return b

        def get_reference_frame(self, frame, device):
            b = self.br[frame % self.k]
            b.copy_(inner.get_reference_frame(frame, device))
            return b

    for k in (1, 3):
        for batch in (None, 4):
            src = Reuser(k)
            mm = fv.fvvdp(display_name="standard_fhd", batch_frames=batch)
            q1, s1 = mm.predict_video_source(src)
            assert abs(float(q1) - float(oq)) < 2e-5, (k, batch, float(q1), float(oq))
            assert np.array_equal(s1["Q_per_ch"], s0["Q_per_ch"]) or np.allclose(s1["Q_per_ch"], s0["Q_per_ch"], rtol=2e-4, atol=1e-6 * s0["Q_per_ch"].max())
            assert src.n <= N + k + 1 + 8 + 7              # every frame once, plus the ones the reuse overwrote and (when the
                                                           # reuse shows after the first batch) that batch's window again
    q2, s2 = m.predict_video_source(Reuser(1), sync=False)
    fv.fvvdp.finish(s2)
    assert abs(float(q2) - float(q0)) < 5e-6


def test_source_whose_buffer_ring_is_shorter_than_window_plus_batch(fv):
    """a ring of R buffers with 2 < R <= window: the reuse only shows when a later batch is fetched, and the frames it
    overwrote belong to the PREVIOUS batch, whose kernels may still be reading them on the caller's stream (the fetches run on a
    side stream).  The feeder must not only repair the batch being fetched but have the previous batch evaluated again.  60 fps
    (15 taps), default schedule (batches of 8, 16, ...), R = 16 and 20, frames large enough that the previous batch is still
    running when the source overwrites its input.  Also a source that returns ONE buffer for the test and the reference frame."""
    from fovvideovdp_amd.synth import synth_video_pair
    N, H, W, fps = 44, 540, 960, 60
    test, ref = synth_video_pair(N, H, W, device="cuda")
    m = fv.fvvdp(display_name="standard_fhd")
    q0, s0 = m.predict(test, ref, frames_per_second=fps)
    inner = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)

    class Ring(fv.fvvdp_video_source):
        def __init__(self, k, shared=False):
            self.k, self.shared = k, shared
            self.bt = [torch.empty((1, 1, 1, H, W), device="cuda") for _ in range(k)]
            self.br = self.bt if shared else [torch.empty((1, 1, 1, H, W), device="cuda") for _ in range(k)]

        def get_video_size(self):
            return (H, W, N)

        def get_frames_per_second(self):
            return fps

        def get_test_frame(self, frame, device):
            b = self.bt[frame % self.k]
            b.copy_(inner.get_test_frame(frame, device))
            return b

        def get_reference_frame(self, frame, device):
            b = self.br[frame % self.k]
            b.copy_(inner.get_reference_frame(frame, device))
            return b

    for src in (Ring(16), Ring(20), Ring(3), Ring(64, shared=True), Ring(1, shared=True)):
        for rep in range(3):                                  # a race does not lose every time
            q1, s1 = fv.fvvdp(display_name="standard_fhd").predict_video_source(src)
            assert abs(float(q1) - float(q0)) < 5e-6, (src.k, src.shared, rep, float(q1), float(q0))
            a, b = s1["Q_per_ch"].astype(np.float64), s0["Q_per_ch"].astype(np.float64)
            assert np.all(np.abs(a - b) <= 2e-4 * np.abs(b) + 1e-6 * np.max(b)), (src.k, src.shared, rep)


def test_frames_produced_on_the_callers_stream_just_before_the_call(fv):
    """the feeder's side stream waits for the caller's current stream.  The clip is generated on the GPU by a long
    chain of kernels queued right before predict; without the wait the side stream reads it half-written."""
    from fovvideovdp_amd.synth import synth_video_pair
    N, H, W, fps = 12, 270, 480, 30
    test, ref = synth_video_pair(N, H, W, device="cuda")
    m = fv.fvvdp(display_name="standard_fhd")
    lt = m.display_photometry.forward(test.float() / 255)
    lr = m.display_photometry.forward(ref.float() / 255)
    w = torch.tensor([0.2126, 0.7152, 0.0722], device="cuda").view(1, 3, 1, 1, 1)
    Lt0, Lr0 = (lt * w).sum(1, keepdim=True), (lr * w).sum(1, keepdim=True)

    class Lum(fv.fvvdp_video_source):
        def __init__(self, a, b):
            self.a, self.b = a, b

        def get_video_size(self):
            return (H, W, N)

        def get_frames_per_second(self):
            return fps

        def get_test_frame(self, frame, device):
            return self.a[:, :, frame:frame + 1]

        def get_reference_frame(self, frame, device):
            return self.b[:, :, frame:frame + 1]

    q_ref, _ = m.predict_video_source(Lum(Lt0.clone(), Lr0.clone()))
    torch.cuda.synchronize()
    for _ in range(3):
        A = torch.zeros_like(Lt0)
        B = torch.zeros_like(Lr0)
        junk = torch.randn(4096, 4096, device="cuda")
        for _ in range(6):                                # ~ms of queued work ahead of the writes below
            junk = junk @ junk * 1e-4
        A.copy_(Lt0 + 0.0 * junk[0, 0])
        B.copy_(Lr0 + 0.0 * junk[0, 0])
        q, _ = m.predict_video_source(Lum(A, B))          # no synchronisation in between
        assert float(q) == float(q_ref)


def test_frame_sharded_call_uploads_only_its_own_frames(fv):
    """VERDICT r2 item 5 / SURVEY 8(e): with host-resident arrays a frame-sharded call moves its own frames plus the fl-1
    frames of temporal halo over PCIe, not the clip; results are bit-equal to the unsharded call on resident arrays."""
    from fovvideovdp_amd.synth import synth_video_pair
    from fovvideovdp_amd.sharding import shard_range
    from oracle import fvvdp_oracle as orc
    N, H, W, fps, world = 64, 72, 128, 30, 8
    fl = 8
    test, ref = synth_video_pair(N, H, W)                 # host tensors
    frame_bytes = 2 * 3 * H * W                           # both streams, uint8 RGB
    m = fv.fvvdp(display_name="standard_fhd")
    q_all, s_all = m.predict(test.cuda(), ref.cuda(), frames_per_second=fps)
    assert m.last_h2d_bytes == 0
    vs = fv.fvvdp_video_source_array(test, ref, fps, display_photometry=m.display_photometry)
    Q = np.zeros_like(s_all["Q_per_ch"])
    for pad in ("replicate", "circular"):
        mm = fv.fvvdp(display_name="standard_fhd", temp_padding=pad)
        qp, sp = mm.predict(test.cuda(), ref.cuda(), frames_per_second=fps)
        for rank in range(world):
            f0, f1 = shard_range(N, rank, world)
            _, st = mm.predict_video_source(vs, frame_range=(f0, f1), pool=False)
            share = (f1 - f0) * frame_bytes
            halo = (fl - 1) * frame_bytes
            assert mm.last_h2d_bytes <= 1.2 * share + halo, (pad, rank, mm.last_h2d_bytes, share, halo)
            assert mm.last_h2d_bytes >= share
            Q[:, :, f0:f1] = st["Q_per_ch"]
        # the same per-pixel work; the chunk heights (grouping of the fp32 partial sums) follow the batch size
        a, b = Q.astype(np.float64), sp["Q_per_ch"].astype(np.float64)
        assert np.all(np.abs(a - b) <= 3e-6 * np.abs(b) + 1e-9 * np.max(b)), pad
    oq, ost = orc.Oracle("standard_fhd").predict(test.numpy(), ref.numpy(), frames_per_second=fps, frames=range(0, 10))
    o = ost["Q_per_ch"].astype(np.float64)
    assert np.all(np.abs(s_all["Q_per_ch"][:, :, :10] - o) <= 1e-3 * np.abs(o) + 1e-6 * np.max(o))


def test_chunk_mapped_scratch_gives_the_same_results_as_hipmalloc(tmp_path):
    """The large pyramid levels are mapped from 32 MB physical chunks through the virtual-memory API (the temporal kernel is 15 %
    faster than on a physically contiguous hipMalloc range, profiles/r04_level0_chunks.md); FVVDP_ALLOC=malloc (read once per
    process) goes back to hipMalloc.  Same bits either way, also for another chunk size and across context re-creation."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import fovvideovdp_amd as fv\n"
        "from fovvideovdp_amd.synth import synth_video_pair\n"
        "t, r = synth_video_pair(40, 1080, 1920, device='cuda')\n"         # level 0: 40 x 33 MB = 1.3 GB, level 1 0.33 GB: both mapped
        "out = []\n"
        "for rep in range(2):\n"
        "    m = fv.fvvdp(display_name='standard_fhd')\n"
        "    for k in range(2):\n"
        "        q, st = m.predict(t, r, frames_per_second=30)\n"
        "        out.append(np.concatenate([st['Q_per_ch'].reshape(-1), [float(q)]]))\n"
        "    free0 = torch.cuda.mem_get_info()[0]\n"
        "    del m\n"
        "    import gc; gc.collect()\n"
        "    print('FREED', torch.cuda.mem_get_info()[0] - free0)\n"
        "np.save(sys.argv[1], np.stack(out))\n" % root)
    res = {}
    for mode, env in (("vmm", {}), ("vmm8", {"FVVDP_VMM_CHUNK_MB": "8"}), ("malloc", {"FVVDP_ALLOC": "malloc"})):
        f = str(tmp_path / ("a_%s.npy" % mode))
        p = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        res[mode] = np.load(f)
        freed = [int(l.split()[1]) for l in p.stdout.splitlines() if l.startswith("FREED")]
        assert len(freed) == 2 and min(freed) > 1.5e9, (mode, freed)      # destroying the context returns the mapped chunks
    assert all(np.array_equal(res["vmm"][k], res["vmm"][0]) for k in range(4))
    assert np.array_equal(res["vmm"], res["malloc"]) and np.array_equal(res["vmm"], res["vmm8"])


def test_level0_in_two_chosen_ranges_changes_no_bits(tmp_path):
    """Level 0 of a video context that holds >= 1 GiB lives in TWO ranges (even / odd frame slots) that fvvdp_ctx_create chooses among N
    half-size candidates of two kinds by a streaming-write probe over all pairs (fvvdp_ctx_alloc_info: state 9 from the first call on,
    the kinds of the two ranges, the highest / lowest pair rate, the layout's timing); the other candidates are freed before the first user
    call.  Per-frame calls never allocate, free or synchronise (fvvdp_ctx_call_stats: all zero from the FIRST call on -- SURVEY 8(b)
    "allocated once in ctx_create").  Same bits with one range (choice off), with hipMalloc candidates only, with 2 or 4
    candidates, and with the further candidates that are taken while no pair reaches the rate of two different classes (forced by an
    unreachable rate: all of them tried, the memory given back); small contexts and images keep their one range."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, ctypes as C, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import fovvideovdp_amd as fv\n"
        "from fovvideovdp_amd import _native as nat\n"
        "from fovvideovdp_amd.synth import synth_video_pair\n"
        "def info(m):\n"
        "    st, cm, n, kept = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(-1); us = (C.c_float * 8)()\n"
        "    nat.check(nat.lib().fvvdp_ctx_alloc_info(m._ctx.handle, C.byref(st), C.byref(cm), us, 8, C.byref(n), C.byref(kept)))\n"
        "    cs = (C.c_int64 * 3)()\n"
        "    nat.check(nat.lib().fvvdp_ctx_call_stats(m._ctx.handle, cs))\n"
        "    return [st.value, cm.value, n.value, kept.value, cs[0], cs[1], cs[2]] + [round(us[k], 2) for k in range(8)]\n"
        "t, r = synth_video_pair(41, 1080, 1920, device='cuda')\n"
        "torch.cuda.synchronize()\n"
        "free0 = torch.cuda.mem_get_info()[0]\n"
        "m = fv.fvvdp(display_name='standard_fhd')\n"
        "out = []\n"
        "for k in range(4):\n"
        "    q, st = m.predict(t, r, frames_per_second=30)\n"
        "    out.append(np.concatenate([st['Q_per_ch'].reshape(-1), [float(q)]]))\n"
        "    print('INFO', k, *info(m))\n"
        "scratch = nat.lib().fvvdp_ctx_scratch_bytes(m._ctx.handle)\n"
        "print('MEM', free0 - torch.cuda.mem_get_info()[0], scratch)\n"
        "m2 = fv.fvvdp(display_name='standard_fhd')\n"
        "t2, r2 = synth_video_pair(20, 270, 480, device='cuda')\n"
        "for k in range(3): m2.predict(t2, r2, frames_per_second=30)\n"
        "print('SMALL', *info(m2))\n"
        "np.save(sys.argv[1], np.stack(out))\n" % root)
    res = {}
    modes = (("default", {}, 6), ("off", {"FVVDP_PLACEMENT_PROBE": "0"}, 0), ("malloc", {"FVVDP_ALLOC": "malloc"}, 6),
             ("further", {"FVVDP_PLACEMENT_MIXED_TBS": "99", "FVVDP_PLACEMENT_EXTRA": "5"}, 6),      # no pair is ever good enough: all five further candidates are tried
             ("two", {"FVVDP_PLACEMENT_PROBE": "2"}, 2), ("four", {"FVVDP_PLACEMENT_PROBE": "4"}, 4))
    for mode, env, n_want in modes:
        f = str(tmp_path / ("s_%s.npy" % mode))
        p = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=400)
        assert p.returncode == 0, p.stderr[-1500:]
        res[mode] = np.load(f)
        infos = [l.split()[2:] for l in p.stdout.splitlines() if l.startswith("INFO")]
        small = [l.split()[1:] for l in p.stdout.splitlines() if l.startswith("SMALL")][0]
        assert int(small[0]) == 9 and int(small[1]) < 100 and int(small[2]) == 0 and int(small[3]) == -1      # a 130 MB level 0 stays one range
        assert [int(v) for v in small[4:7]] == [0, 0, 0]
        assert len(infos) == 4
        for i in infos:
            state, kind, n, kept = (int(v) for v in i[:4])
            assert state == 9 and n == n_want, (mode, i)
            assert [int(v) for v in i[4:7]] == [0, 0, 0], (mode, i)                      # no sync / alloc / free in per-frame calls
            us = [float(v) for v in i[7:]]
            if n_want:
                lo, hi = kept % 8, kept // 8
                assert 0 <= lo < hi < n_want and kind >= 100, (mode, i)
                assert us[0] == 0 and us[1] >= us[2] > 0 and all(u == 0 for u in us[4:]), (mode, i)      # us[0]: the creation-time timing pass of round 5 is gone
                assert us[3] == (5 if mode == "further" else us[3]) and 0 <= us[3] <= 8, (mode, i)
                want = {"malloc": [0] * 6}.get(mode, [1, 0, 1, 0, 1, 0])    # candidate kinds: chunk-mapped and hipMalloc in turn
                if us[3] == 0:
                    assert kind == 100 + 10 * want[hi] + want[lo], (mode, i)
            else:
                assert kept == -1 and kind == 1
        assert infos[0] == infos[3]                                                       # settled at creation: nothing moves afterwards
        used, scratch = [int(v) for v in [l for l in p.stdout.splitlines() if l.startswith("MEM")][0].split()[1:]]
        assert used < scratch + (700 << 20), (mode, used, scratch)                       # the other candidates and the synthetic clip are gone
    for mode in res:
        assert all(np.array_equal(res[mode][k], res[mode][0]) for k in range(4)), mode
        assert np.array_equal(res[mode], res["default"]), mode


def test_level0_in_two_ranges_every_path_same_bits(monkeypatch):
    """FVVDP_LEVEL0_SPLIT=1 puts level 0 of ANY context in two ranges (even / odd frame slots; L0Addr in every kernel that touches level
    0) without a choice: every writer and reader of level 0 -- the vector, per-pixel, generic, two-pass and YUV temporal kernels, the
    planar hand-over and export, both pyramid kernels incl. the foveated and the map-writing variants, the colouring of heat maps --
    gives the bits of the one-range layout, with odd and even frame counts and with slot offsets."""
    import ctypes as C
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.synth import synth_video_pair, synth_yuv_pair, synth_gaze
    from lowlevel import Pipeline

    def kind_of(m):
        st, cm, n, kept = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(-1)
        us = (C.c_float * 8)()
        nat.check(nat.lib().fvvdp_ctx_alloc_info(m._ctx.handle, C.byref(st), C.byref(cm), us, 8, C.byref(n), C.byref(kept)))
        return cm.value
    t, r = synth_video_pair(11, 136, 244, device="cuda")
    to, ro = synth_video_pair(6, 135, 241, device="cuda")                    # pixel count not a multiple of 4: per-pixel ring kernel
    ty, ry = synth_yuv_pair(10, 136, 244, 8, "420", device="cuda")
    gaze = synth_gaze(11, 136, 244).numpy()
    cases = [
        ("video u8", dict(display_name="standard_fhd"), lambda m: m.predict(t, r, frames_per_second=30)),
        ("video u8 60 fps", dict(display_name="standard_fhd"), lambda m: m.predict(t, r, frames_per_second=60)),
        ("video f32", dict(display_name="standard_fhd"), lambda m: m.predict(t.float() / 255, r.float() / 255, frames_per_second=30)),
        ("video u16 gray", dict(display_name="standard_hdr_pq"), lambda m: m.predict((t[:, 1:2].to(torch.int32) * 257).to(torch.int16), (r[:, 1:2].to(torch.int32) * 257).to(torch.int16), frames_per_second=30)),
        ("video 240 fps (two-pass path for float input)", dict(display_name="standard_fhd"), lambda m: m.predict(t.float() / 255, r.float() / 255, frames_per_second=240)),
        ("odd size", dict(display_name="standard_fhd"), lambda m: m.predict(to, ro, frames_per_second=30)),
        ("sub-batches", dict(display_name="standard_fhd", batch_frames=4), lambda m: m.predict(t, r, frames_per_second=30)),
        ("image", dict(display_name="standard_4k"), lambda m: m.predict(t[0, :, 0].permute(1, 2, 0), r[0, :, 0].permute(1, 2, 0), dim_order="HWC")),
        ("foveated", dict(display_name="standard_hdr_pq", foveated=True), lambda m: m.predict(t, r, frames_per_second=30, fixation_point=gaze)),
        ("heat map", dict(display_name="standard_fhd", heatmap="threshold"), lambda m: m.predict(t, r, frames_per_second=30)),
        ("yuv", dict(display_name="standard_fhd"), lambda m: m.predict_video_source(
            fv.fvvdp_video_source_yuv_frames(ty, ry, 30, 244, 136, bit_depth=8, chroma_ss="420", display_photometry=m.display_photometry))),
    ]
    out = {}
    R = torch.rand((7, 4, 136, 244), device="cuda") * 100 + 1
    for split in ("0", "1"):
        monkeypatch.setenv("FVVDP_LEVEL0_SPLIT", split)
        for name, kw, call in cases:
            m = fv.fvvdp(**kw)
            q, st = call(m)
            assert (kind_of(m) >= 100) == (split == "1"), (name, split, kind_of(m))
            res = [st["Q_per_ch"].copy(), np.float64(float(q))]
            if "heatmap" in st:
                res.append(st["heatmap"].clone().numpy())
            out[(name, split)] = res
        # the planar hand-over and the export of level 0 through the C ABI, slots 3 .. 9 of a 10-slot context
        m = fv.fvvdp(display_name="standard_fhd")
        pl = Pipeline(m, 244, 136, 4, 10)
        pl.load_planar(R, slot0=3)
        pl.load_planar(R[:3] * 0.5, slot0=0)
        back = pl.export_level(0, 10)
        assert torch.equal(back[3:], R) and torch.equal(back[:3], R[:3] * 0.5), split
        out[("planar", split)] = [pl.bands_forward(10).cpu().numpy()]
        pl.close()
    differ = [name for name in [c[0] for c in cases] + ["planar"]
              if not (len(out[(name, "0")]) == len(out[(name, "1")]) and all(np.array_equal(x, y) for x, y in zip(out[(name, "0")], out[(name, "1")])))]
    assert not differ, differ


def test_per_frame_calls_do_not_sync_or_allocate_optional_paths_only_on_first_use():
    """fvvdp_ctx_call_stats over the optional paths: the foveated tables are built in the first call of a context (one sync, their
    allocations) and never again while the geometry stays; 60 fps (16-slot ring + block counter), uint16 and float sources: nothing."""
    import ctypes as C
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.synth import synth_video_pair

    def stats(m):
        cs = (C.c_int64 * 3)()
        nat.check(nat.lib().fvvdp_ctx_call_stats(m._ctx.handle, cs))
        return [int(cs[0]), int(cs[1]), int(cs[2])]
    t, r = synth_video_pair(12, 270, 480, device="cuda")
    for kw, src in ((dict(frames_per_second=60), (t, r)), (dict(frames_per_second=30), ((t.to(torch.int16) * 257), (r.to(torch.int16) * 257))),
                    (dict(frames_per_second=30), (t.float() / 255, r.float() / 255)), (dict(frames_per_second=30, dim_order="BCFHW"), (t[:, 1:2], r[:, 1:2]))):
        m = fv.fvvdp(display_name="standard_fhd")
        for rep in range(3):
            m.predict(src[0], src[1], **kw)
            assert stats(m) == [0, 0, 0], (kw, rep)
    mf = fv.fvvdp(display_name="standard_fhd", foveated=True)
    mf.predict(t, r, frames_per_second=30, fixation_point=[100, 60])
    first = stats(mf)
    assert first[0] == 1 and first[1] >= 1 and first[2] == 0, first
    for rep in range(3):
        mf.predict(t, r, frames_per_second=30, fixation_point=[200 + rep, 90])
        assert stats(mf) == first


def test_ticketed_temporal_kernel_changes_no_bits(monkeypatch):
    """uint8 sources with 9..16 taps (33-64 fps) run the temporal kernel as resident workgroups that take their pixel blocks from a
    counter (more than two rounds of blocks only; FVVDP_K1_TICKET=0 = one workgroup per block).  Same blocks, same arithmetic: the
    results are bit-identical, RGB and gray, for a frame whose last block is partial, and call after call (the counter is zeroed
    before every launch)."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    for (N, H, W, fps) in ((20, 2160, 3840, 60), (10, 1442, 2564, 50), (9, 1080, 1920, 60)):
        t, r = synth_video_pair(N, H, W, device="cuda")
        for gray in (False, True):
            tt, rr = (t[:, :1], r[:, :1]) if gray else (t, r)
            out = {}
            for mode in ("1", "0"):
                monkeypatch.setenv("FVVDP_K1_TICKET", mode)
                m = fv.fvvdp(display_name="standard_4k")
                res = []
                for rep in range(3):
                    q, st = m.predict(tt, rr, frames_per_second=fps)
                    res.append(np.concatenate([st["Q_per_ch"].reshape(-1), [float(q)]]))
                out[mode] = np.stack(res)
            assert np.array_equal(out["1"], out["0"]), (N, H, W, fps, gray)
            assert np.array_equal(out["1"][0], out["1"][2])


def test_context_made_for_a_longer_clip_serves_shorter_ones(fv):
    """Clips of different lengths through one metric object: the native context made for the longer clip (scratch for its batch, the
    level-0 ranges chosen at its creation) serves the shorter ones -- no second context; a longer clip replaces it.  Same bits as a
    fresh object per clip."""
    from fovvideovdp_amd.synth import synth_video_pair
    t, r = synth_video_pair(24, 96, 160, device="cuda")
    m = fv.fvvdp(display_name="standard_fhd")
    q24, s24 = m.predict(t, r, frames_per_second=30)
    ctx = m._ctx
    for n in (13, 24, 7, 2):
        q, s = m.predict(t[:, :, :n], r[:, :, :n], frames_per_second=30)              # [1, 3, F, H, W]
        assert m._ctx is ctx, n
        qf, sf = fv.fvvdp(display_name="standard_fhd").predict(t[:, :, :n], r[:, :, :n], frames_per_second=30)
        assert float(q) == float(qf) and np.array_equal(s["Q_per_ch"], sf["Q_per_ch"]), n
    t2, r2 = synth_video_pair(31, 96, 160, device="cuda")
    q31, s31 = m.predict(t2, r2, frames_per_second=30)
    assert m._ctx is not ctx and m._ctx.key[4] == 31
    qf, sf = fv.fvvdp(display_name="standard_fhd").predict(t2, r2, frames_per_second=30)
    assert float(q31) == float(qf) and np.array_equal(s31["Q_per_ch"], sf["Q_per_ch"])
    # a different frame size never reuses it
    t3, r3 = synth_video_pair(5, 100, 160, device="cuda")
    c31 = m._ctx
    m.predict(t3, r3, frames_per_second=30)
    assert m._ctx is not c31
