"""The C ABI with plain device pointers only: device memory from the HIP runtime through ctypes (hipMalloc /
hipMemcpy / hipStreamCreate), no torch tensor anywhere near the library.  The whole video path -- temporal channels from
uint8 RGB frames, pyramid + CSF + masking + pooling, JOD regression -- runs through include/fvvdp_hip.h alone and is
checked against the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hip():
    h = C.CDLL("libamdhip64.so")
    h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    h.hipFree.argtypes = [C.c_void_p]
    h.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    h.hipStreamSynchronize.argtypes = [C.c_void_p]
    h.hipStreamDestroy.argtypes = [C.c_void_p]
    return h


H2D, D2H = 1, 2


def test_video_path_through_the_c_abi_only():
    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.fvvdp import band_frequencies, window_frame_indices
    from fovvideovdp_amd.synth import synth_video_pair
    from oracle import fvvdp_oracle as orc
    hip, lib = _hip(), nat.lib()
    N, H, W, fps = 9, 54, 96, 30
    test, ref = synth_video_pair(N, H, W)                          # CPU tensors, used as plain host bytes below
    t_np, r_np = np.ascontiguousarray(test.numpy()[0]), np.ascontiguousarray(ref.numpy()[0])      # [3, N, H, W] uint8

    # host-side preparation (display model, filter taps, CSF tables): the metric object on the CPU, no device work
    m = fv.fvvdp(display_name="standard_fhd", device="cpu")
    n_bands, rho_band = band_frequencies(W, H, m.pix_per_deg)
    fl = int(np.ceil(250.0 / (1000.0 / fps)))
    m.filter_len = fl                                             # like predict_video_source (fvvdp.py:228-230)
    F, _ = m.get_temporal_filters(fps)
    taps = np.ascontiguousarray(F.numpy(), dtype=np.float32)
    idx = np.ascontiguousarray(window_frame_indices(N, fl, "replicate"), dtype=np.int32)
    import torch
    codes = torch.arange(256, dtype=torch.float32) / 255.0
    lut = m.display_photometry.forward(codes.view(1, 1, 1, 1, 256)).reshape(-1).numpy().astype(np.float32)
    y_log, tab = m.csf_tables_1d(rho_band, n_bands)
    rgb2y = np.asarray(fv.fvvdp_video_source_array(test, ref, fps, display_photometry="standard_fhd").color_to_luminance,
                       dtype=np.float32)

    def dev(nbytes, host=None):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), nbytes) == 0
        if host is not None:
            assert hip.hipMemcpy(p, host.ctypes.data_as(C.c_void_p), nbytes, H2D) == 0
        else:
            assert hip.hipMemset(p, 0, nbytes) == 0
        return p

    stream = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    d_t, d_r = dev(t_np.nbytes, t_np), dev(r_np.nbytes, r_np)
    d_lut = dev(lut.nbytes, lut)
    d_Q, d_jod, d_flag = dev(n_bands * 2 * N * 4), dev(4), dev(4)
    ctx = C.c_void_p()
    rb = (C.c_double * (n_bands + 1))(*[float(x) for x in rho_band])
    prm = m.native_params()
    try:
        nat.check(lib.fvvdp_ctx_create(C.byref(ctx), W, H, n_bands, 4, N, rb, C.byref(prm)))
        nat.check(lib.fvvdp_ctx_set_csf_1d(ctx, nat.fptr(y_log), nat.fptr(tab)))
        e = nat.Eotf()
        e.kind, e.d_lut = nat.EOTF_LUT, d_lut.value
        nat.check(lib.fvvdp_temporal_channels(ctx, d_t, d_r, nat.FVVDP_U8, 3, N * H * W, H * W, C.byref(e), nat.fptr(rgb2y),
                                              idx.ctypes.data_as(C.POINTER(C.c_int32)), nat.fptr(taps), fl, N, 0, d_flag, stream))
        nat.check(lib.fvvdp_bands_forward(ctx, N, d_Q, N, 0, None, None, None, stream))
        pp = nat.PoolParams(m.beta_sch, m.beta_tch, m.beta_t, m.w_transient, m.jod_a, float(10.0 ** m.log_jod_exp))
        nat.check(lib.fvvdp_pool_jod(d_Q, n_bands, 2, N, N, C.byref(pp), d_jod, stream))
        assert hip.hipStreamSynchronize(stream) == 0
        Q = np.zeros((n_bands, 2, N), dtype=np.float32)
        jod = np.zeros(1, dtype=np.float32)
        flag = np.zeros(1, dtype=np.int32)
        assert hip.hipMemcpy(Q.ctypes.data_as(C.c_void_p), d_Q, Q.nbytes, D2H) == 0
        assert hip.hipMemcpy(jod.ctypes.data_as(C.c_void_p), d_jod, 4, D2H) == 0
        assert hip.hipMemcpy(flag.ctypes.data_as(C.c_void_p), d_flag, 4, D2H) == 0
        # the same two stages through the combined entry point: bit-identical JOD and Q
        assert hip.hipMemset(d_jod, 0, 4) == 0
        nat.check(lib.fvvdp_bands_forward_pool(ctx, N, d_Q, N, 0, None, None, None, C.byref(pp), d_jod, stream))
        assert hip.hipStreamSynchronize(stream) == 0
        Q2 = np.zeros_like(Q)
        jod2 = np.zeros(1, dtype=np.float32)
        assert hip.hipMemcpy(Q2.ctypes.data_as(C.c_void_p), d_Q, Q2.nbytes, D2H) == 0
        assert hip.hipMemcpy(jod2.ctypes.data_as(C.c_void_p), d_jod, 4, D2H) == 0
        assert np.array_equal(Q2, Q) and jod2[0] == jod[0]
    finally:
        if ctx:
            lib.fvvdp_ctx_destroy(ctx)
        for p in (d_t, d_r, d_lut, d_Q, d_jod, d_flag):
            hip.hipFree(p)
        hip.hipStreamDestroy(stream)
    # the raw-pointer path IS the product path: the same pair through the Python API (torch tensors, the caching allocator's memory, torch's
    # stream) gives the same bits (the 2e-3 against the oracle below proves that the ABI is drivable, this that
    # nothing differs between the two ways in)
    q_api, st_api = fv.fvvdp(display_name="standard_fhd").predict(test, ref, frames_per_second=fps)
    assert np.array_equal(st_api["Q_per_ch"], Q) and float(q_api) == float(jod[0])
    oq, ost = orc.Oracle("standard_fhd").predict(test.numpy(), ref.numpy(), frames_per_second=fps)
    assert flag[0] == 0
    assert abs(float(jod[0]) - float(oq)) < 1e-4
    a, b = Q.astype(np.float64), ost["Q_per_ch"].astype(np.float64)
    assert np.all(np.abs(a - b) <= 2e-3 * np.abs(b) + 1e-5 * np.max(b))
