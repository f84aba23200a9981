"""Test helper (not part of the product): thin object wrapper over the C ABI for tests that drive the stages themselves.
All tensors are torch CUDA tensors; work is enqueued on torch's current stream."""
import ctypes as C

import numpy as np
import torch

from fovvideovdp_amd import _native as nat


class Pipeline:
    def __init__(self, metric, width, height, planes, max_frames, foveated=False):
        from fovvideovdp_amd.fvvdp import band_frequencies
        self.m = metric
        self.W, self.H, self.P, self.max_frames = width, height, planes, max_frames
        self.n_bands, self.rho_band = band_frequencies(width, height, metric.pix_per_deg)
        self.dev = metric.device
        self.lib = nat.lib()
        self.handle = C.c_void_p()
        rb = (C.c_double * (self.n_bands + 1))(*[float(r) for r in self.rho_band])
        prm = metric.native_params()
        with torch.cuda.device(self.dev):
            nat.check(self.lib.fvvdp_ctx_create(C.byref(self.handle), width, height, self.n_bands, planes, max_frames,
                                                rb, C.byref(prm)))
            if foveated:
                for cc in range(2):
                    l = metric.csf_lut[cc]
                    nat.check(self.lib.fvvdp_ctx_set_csf_3d(self.handle, cc, nat.fptr(l["S_log"]), nat.fptr(l["Y_log"]),
                                                            nat.fptr(l["rho_log"]), nat.fptr(l["ecc_sqrt"])))
            else:
                y_log, tab = metric.csf_tables_1d(self.rho_band, self.n_bands)
                nat.check(self.lib.fvvdp_ctx_set_csf_1d(self.handle, nat.fptr(y_log), nat.fptr(tab)))
        self.foveated = foveated

    def close(self):
        if self.handle:
            self.lib.fvvdp_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        nat.check(self.lib.fvvdp_ctx_level_size(self.handle, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def scratch_bytes(self):
        return int(self.lib.fvvdp_ctx_scratch_bytes(self.handle))

    def load_planar(self, R, slot0=0):
        """R: [n,P,H,W] fp32 CUDA tensor in the reference's plane order."""
        R = R.contiguous()
        assert R.shape[1:] == (self.P, self.H, self.W) and R.dtype == torch.float32
        nat.check(self.lib.fvvdp_load_channels_planar(self.handle, C.c_void_p(R.data_ptr()), R.shape[0], slot0, self.stream()))

    def temporal(self, test, ref, dtype, C_ch, chan_stride, frame_stride, eotf, rgb2y, idx, taps, fl, n_out, slot0=0, oob=None):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        w = None if rgb2y is None else nat.fptr(np.ascontiguousarray(rgb2y, dtype=np.float32))
        self._keep = (idx, taps)
        nat.check(self.lib.fvvdp_temporal_channels(
            self.handle, C.c_void_p(test.data_ptr()), C.c_void_p(ref.data_ptr()), dtype, C_ch, chan_stride, frame_stride,
            C.byref(eotf), w, idx.ctypes.data_as(C.POINTER(C.c_int32)), nat.fptr(taps), fl, n_out, slot0,
            C.c_void_p(oob.data_ptr()) if oob is not None else None, self.stream()))

    def export_level(self, level, n):
        w, h = self.level_size(level)
        out = torch.empty((n, self.P, h, w), dtype=torch.float32, device=self.dev)
        nat.check(self.lib.fvvdp_export_level(self.handle, level, n, C.c_void_p(out.data_ptr()), self.stream()))
        return out

    def bands_forward(self, n, want_maps=False, fixation=None):
        """Returns Q[n_bands,2,n] (CUDA) and, if want_maps, a list of per-band dicts D/contrast/lbkg/S."""
        Q = torch.zeros((self.n_bands, 2, n), dtype=torch.float32, device=self.dev)
        maps_arr, maps = None, None
        if want_maps:
            maps_arr = (nat.BandMaps * self.n_bands)()
            maps = []
            for b in range(self.n_bands):
                w, h = self.level_size(b)
                d = dict(D=torch.zeros((n, 2, h, w), device=self.dev), contrast=torch.zeros((n, self.P, h, w), device=self.dev),
                         lbkg=torch.zeros((n, h, w), device=self.dev), S=torch.zeros((n, 2, h, w), device=self.dev))
                maps_arr[b].d_D, maps_arr[b].d_contrast = d["D"].data_ptr(), d["contrast"].data_ptr()
                maps_arr[b].d_lbkg, maps_arr[b].d_S = d["lbkg"].data_ptr(), d["S"].data_ptr()
                maps.append(d)
        fx, g = None, None
        if fixation is not None:
            fxa = np.ascontiguousarray(fixation, dtype=np.float32).reshape(n, 2)
            fx = nat.fptr(fxa)
            g = C.byref(self.m._geom_struct())
        nat.check(self.lib.fvvdp_bands_forward(self.handle, n, C.c_void_p(Q.data_ptr()), n, 0, fx, g, maps_arr, self.stream()))
        return (Q, maps) if want_maps else Q

    def timing_enable(self, on=True):
        nat.check(self.lib.fvvdp_ctx_timing_enable(self.handle, 1 if on else 0))

    def timing_read(self, reset=True):
        nk = self.n_bands + 2
        ms = (C.c_float * nk)()
        cnt = (C.c_int32 * nk)()
        nat.check(self.lib.fvvdp_ctx_timing_read(self.handle, ms, cnt, nk, 1 if reset else 0))
        return np.array(ms[:], dtype=np.float64), np.array(cnt[:], dtype=np.int64)
