"""`class fvvdp`: the reference's public metric API (pyfvvdp/fvvdp.py:58) on top of the MI355X HIP hot path.

What stays in Python (as in the reference): configuration, display models, temporal-filter taps, the collapse of
the CSF look-up table to per-band 1-D tables, frame batching, and the final pooling / JOD regression
(do_pooling_and_jods).  What runs in hand-written HIP kernels (libfvvdp_hip.so, include/fvvdp_hip.h): unpacking +
display photometry + luminance + temporal filtering, the Gaussian/contrast pyramid, CSF weighting, mutual masking
and spatial pooling.  There is no CPU implementation of the hot path in this package: without a GPU or without the
built library `predict*` raises.
"""
import ctypes as C
import json
import logging
import math

import numpy as np
import torch

from . import _native as nat
from . import utils
from .display_model import (code_value_tables, fvvdp_display_geometry, fvvdp_display_photometry, native_eotf,
                            native_geometry)
from .video_source import fvvdp_video_source_array
from .video_source_yuv import fvvdp_video_source_yuv_frames


def _interpolants(x_q, x):
    """Bracketing knots and fraction of the reference's LUT interpolation, incl. its +1e-6 in the denominator."""
    imax = torch.bucketize(x_q, x)
    imax[imax >= x.shape[0]] = x.shape[0] - 1
    imin = (imax - 1).clamp(0, x.shape[0] - 1)
    ifrc = (x_q - x[imin]) / (x[imax] - x[imin] + 0.000001)
    ifrc[imax == imin] = 0.
    ifrc[ifrc < 0.0] = 0.
    return imin, imax, ifrc


_BAND_CACHE = {}


def band_frequencies(W, H, ppd):
    """Number of band-pass levels and their centre frequencies [cpd] for a W x H frame at `ppd` pixels/degree
    (same rule as the reference's pyramid constructor, pyfvvdp/fvvdp_lpyr_dec.py:15-49).  Memoised per (W, H, ppd);
    callers get their own copy of the array."""
    key = (int(W), int(H), float(ppd))
    if key not in _BAND_CACHE:
        if len(_BAND_CACHE) > 256:
            _BAND_CACHE.clear()
        _BAND_CACHE[key] = _band_frequencies(W, H, ppd)
    height, freqs = _BAND_CACHE[key]
    return height, freqs.copy()


def _band_frequencies(W, H, ppd):
    max_levels = int(np.floor(np.log2(min(H, W)))) - 1
    octave = 0.3228 * np.power(2.0, -np.arange(0.0, 14.0))
    bands = np.concatenate([[1.0], octave]) * ppd / 2.0
    too_low = np.nonzero(bands <= 0.5)[0]
    max_band = int(too_low[0]) if too_low.size else max_levels
    height = int(np.clip(max_band + 1, 0, max_levels))
    freqs = np.array([1.0] + [0.3228 * 2.0 ** (-f) for f in range(height)]) * ppd / 2.0
    return height, freqs


def window_frame_indices(N, fl, temp_padding):
    """Source frame for every virtual time step: fl-1 history entries (temporal padding before frame 0) followed
    by the newest frame of each of the N outputs.  Matches the window the reference builds at frame 0 and then
    slides (pyfvvdp/fvvdp.py:258-291), including `circular` never showing frame 0 at output 0."""
    if temp_padding == "replicate":
        first = [0] * fl
    elif temp_padding == "circular":
        first = [(N - 1 - fl + kk) % N for kk in range(fl)]
    elif temp_padding == "pingpong":
        seq = list(range(0, N)) + list(range(N - 2, 0, -1))
        hist = []
        while len(hist) < (fl - 1):
            hist = hist + seq
        first = (hist[-(fl - 1):] if fl > 1 else []) + [0]
    else:
        raise RuntimeError('Unknown padding method "{}"'.format(temp_padding))
    return np.asarray(first + list(range(1, N)), dtype=np.int32)


class _Context:
    """Owns one native context (scratch for `batch` frames of a W x H pyramid)."""

    def __init__(self, W, H, height, planes, batch, rho_band, prm):
        self.key = (W, H, height, planes, batch)
        self.handle = C.c_void_p()
        rb = (C.c_double * (height + 1))(*[float(r) for r in rho_band])
        nat.check(nat.lib().fvvdp_ctx_create(C.byref(self.handle), W, H, height, planes, batch, rb, C.byref(prm)))

    def close(self):
        if self.handle:
            nat.lib().fvvdp_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class fvvdp:
    def __init__(self, display_name="standard_4k", display_photometry=None, display_geometry=None, color_space="sRGB",
                 foveated=False, heatmap=None, quiet=False, device=None, temp_padding="replicate",
                 use_checkpoints=False, batch_frames=None):
        assert heatmap in [None, "none", "raw", "threshold", "supra-threshold"], "Unsupported heatmap type"
        assert temp_padding in ["replicate", "circular", "pingpong"], "Unsupported temporal padding method"
        self.quiet = quiet
        self.foveated = foveated
        self.heatmap = heatmap
        self.color_space = color_space
        self.temp_padding = temp_padding
        self.use_checkpoints = use_checkpoints
        self.do_heatmap = (self.heatmap is not None) and (self.heatmap != "none")
        # use_checkpoints only changes how the reference's autograd graph is stored (fvvdp.py:302-304); the HIP path is
        # forward-only, so the flag is accepted and ignored; asking for gradients is what is refused (predict*)
        if device is None:
            device = torch.device('cuda:0') if (torch.cuda.is_available() and torch.cuda.device_count() > 0) else torch.device('cpu')
        self.device = self._indexed(torch.device(device))
        self.batch_frames = batch_frames
        self._ctx = None
        self._lut_dev = code_value_tables()
        self._chan_w = {}
        self._copy_stream = None
        self._filters = {}
        self.timing = None
        self.set_display_model(display_name, display_photometry=display_photometry, display_geometry=display_geometry)
        self.load_config()
        self.csf_cache_dirs = ["csf_cache"]
        self.omega = [0, 5]
        self.csf_lut = [utils.load_csf_lut(om, self.csf_sigma, self.k_cm, self.csf_cache_dirs) for om in self.omega]

    # ---- configuration ------------------------------------------------------------------------------------
    @staticmethod
    def _indexed(device):
        """'cuda' -> 'cuda:<current index>': tensors report an indexed device, comparisons against a bare 'cuda' are never equal
        (resident arrays were counted as uploads and copied device-to-device)."""
        if device.type == "cuda" and device.index is None and torch.cuda.is_available():
            return torch.device("cuda", torch.cuda.current_device())
        return device

    def update_device(self, device):
        self.device = self._indexed(torch.device(device))
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = None
        self._lut_dev.clear()
        self._chan_w = {}
        self._copy_stream = None

    def _drop_context(self):
        """Everything baked into the native context at creation (band frequencies, CSF tables, model constants,
        geometry maps) is stale once the display or the parameters change: the next call builds a fresh one."""
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = None
        self._chan_w = {}                # [1, w_transient] of do_pooling_and_jods: a model parameter like the rest

    def load_config(self):
        parameters = utils.config_files.load("fvvdp_parameters.json")
        self.parameters_file = utils.config_files.find("fvvdp_parameters.json")
        for name in ("mask_p", "mask_c", "pu_dilate", "w_transient", "beta", "beta_t", "beta_tch", "beta_sch",
                     "sustained_sigma", "sustained_beta", "csf_sigma", "sensitivity_correction", "masking_model",
                     "local_adapt", "contrast", "jod_a", "log_jod_exp", "mask_q_sust", "mask_q_trans", "k_cm",
                     "filter_len", "version"):
            setattr(self, name, parameters[name])
        if (self.local_adapt != "gpyr" or self.contrast != "weber" or self.pu_dilate != 0 or
                self.masking_model != "min_mutual_masking_perc_norm2"):
            raise RuntimeError("Only the shipped model variant (local_adapt=gpyr, contrast=weber, pu_dilate=0, "
                               "min_mutual_masking_perc_norm2) is implemented by the HIP path")
        self.debug = False
        self._drop_context()

    def set_display_model(self, display_name="standard_4k", display_photometry=None, display_geometry=None):
        if display_photometry is None:
            self.display_photometry = fvvdp_display_photometry.load(display_name)
            self.display_name = display_name
        else:
            self.display_photometry = display_photometry
            self.display_name = "unspecified"
        if display_geometry is None:
            self.display_geometry = fvvdp_display_geometry.load(display_name)
        else:
            self.display_geometry = display_geometry
        self.pix_per_deg = self.display_geometry.get_ppd()
        self._drop_context()
        self._lut_dev.clear()

    # ---- public prediction API ------------------------------------------------------------------------------
    def predict(self, test_cont, reference_cont, dim_order="BCFHW", frames_per_second=0, fixation_point=None, sync=True):
        vs = fvvdp_video_source_array(test_cont, reference_cont, frames_per_second, dim_order=dim_order,
                                      display_photometry=self.display_photometry, color_space_name=self.color_space)
        return self.predict_video_source(vs, fixation_point=fixation_point, sync=sync)

    def predict_video_source(self, vid_source, fixation_point=None, frame_range=None, pool=True, sync=True):
        """Returns (Q_JOD 0-d tensor on the compute device, stats dict).

        Extensions (not in the reference):
        `frame_range=(f0, f1)` (frame sharding across GPUs) evaluates only output frames [f0, f1); the temporal
        window of frame f0 is still filled from the frames before it.  With `pool=False` the JOD regression is skipped
        and Q_JOD is None (the caller pools after combining shards).
        `sync=False` queues the whole call on the current stream and returns without waiting for the GPU:
        `stats['Q_per_ch']` is then a DEVICE tensor [bands, 2, frames] (not numpy) and the out-of-range flag is left in
        `stats['range_flag']` (int32 device tensor; `fvvdp.finish(stats)` converts both and emits the warning);
        `stats['result_buffer']` is the flat device buffer behind them (Q_per_ch | flag | JOD), the row a multi-GPU caller
        all-reduces.  Many
        pairs can be queued back to back this way (one per call); the context's scratch is reused in stream order.
        """
        if self.device.type != "cuda":
            raise RuntimeError("fovvideovdp_amd needs an AMD GPU (torch device 'cuda'); there is no CPU fallback")
        for name in ("test_video", "reference_video"):
            t = getattr(vid_source, name, None)
            if isinstance(t, torch.Tensor) and t.requires_grad and torch.is_grad_enabled():
                raise RuntimeError("Gradients through the metric are not supported on the HIP path (forward-only kernels); "
                                   "detach the inputs or wrap the call in torch.no_grad()")
        with torch.cuda.device(self.device):       # the library launches on the calling thread's current device
            return self._predict_on_device(vid_source, fixation_point, frame_range, pool, sync)

    def predict_batch(self, pairs, dim_order="BCFHW", frames_per_second=0, fixation_point=None):
        """Extension (BASELINE configs[4]: many independent pairs per GPU): queues every (test, reference) pair of `pairs` on the
        caller's stream without host synchronisation (the context's scratch is reused in stream order).  Returns
        [(Q_JOD, stats)] as `predict(..., sync=False)` does: `stats['Q_per_ch']` are device tensors, `fvvdp.finish(stats)`
        brings one to the host."""
        return [self.predict(t, r, dim_order=dim_order, frames_per_second=frames_per_second, fixation_point=fixation_point,
                             sync=False) for (t, r) in pairs]

    @staticmethod
    def finish(stats):
        """Completes a `sync=False` result in place: device -> host copy of Q_per_ch (one synchronisation) and the
        reference's out-of-range warning."""
        if isinstance(stats.get('Q_per_ch'), torch.Tensor):
            flag = stats.pop('range_flag', None)
            stats.pop('result_buffer', None)
            q = stats['Q_per_ch']
            both = torch.cat([q.reshape(-1), flag.view(torch.float32)]).cpu() if flag is not None else q.reshape(-1).cpu()
            stats['Q_per_ch'] = both[:q.numel()].view(q.shape).numpy()
            if flag is not None and int(both[q.numel():].view(torch.int32)[0]) != 0:
                logging.warning("Pixel outside the valid range 0-1")
        return stats

    def _predict_on_device(self, vid_source, fixation_point, frame_range, pool, sync=True):
        height, width, N_frames = vid_source.get_video_size()
        f0, f1 = (0, N_frames) if frame_range is None else frame_range
        if not (0 <= f0 < f1 <= N_frames):
            raise RuntimeError("frame_range out of bounds")
        n_bands, rho_band = band_frequencies(width, height, self.pix_per_deg)
        if n_bands < 1:
            raise RuntimeError("Frame %dx%d is too small for this display (no band-pass level)" % (width, height))
        is_image = (N_frames == 1)
        planes = 2 if is_image else 4
        if is_image:
            fl, taps = 1, np.ones((2, 1), dtype=np.float32)
        else:
            fps = vid_source.get_frames_per_second()
            self.filter_len = int(np.ceil(250.0 / (1000.0 / fps)))
            fkey = (float(fps), self.filter_len, float(self.sustained_sigma), float(self.sustained_beta))
            if fkey not in self._filters:            # taps depend on the frame rate only: evaluated once per rate
                F, _ = self.get_temporal_filters(fps)
                self._filters[fkey] = (F, np.ascontiguousarray(F.numpy(), dtype=np.float32))
            self.F, taps = self._filters[fkey]
            fl = self.filter_len
            if fl > nat.MAX_TAPS:
                raise RuntimeError("frame rate too high: temporal filter longer than %d taps" % nat.MAX_TAPS)
        fix = self._fixation(fixation_point, width, height, N_frames) if self.foveated else None

        n_out = f1 - f0
        wkey = (N_frames, fl, self.temp_padding, f0, f1, is_image)
        wc = self._filters.get(wkey)
        if wc is None:
            widx = window_frame_indices(N_frames, fl, self.temp_padding) if not is_image else np.zeros(1, np.int32)
            wc = (widx, np.unique(widx[f0:f1 + fl - 1]))
            if len(self._filters) > 64:
                self._filters.clear()
            self._filters[wkey] = wc
        widx, need = wc
        # source frames this call touches: the windows of output frames [f0, f1) (frame sharding: a rank's own frames plus
        # the fl-1 frames of temporal halo before them) -- host-resident sources upload these and nothing else
        self.last_h2d_bytes = 0
        feeder = self._make_feeder(vid_source, width, height, need)
        batch = self._batch_size(width, height, planes, n_out, fl)
        schedule = None
        if self.batch_frames is None and getattr(feeder, "preferred_batch", None):
            batch = max(1, min(batch, feeder.preferred_batch))
            if not self.do_heatmap:
                schedule = feeder.batch_schedule(n_out, batch)
        heatmap = None
        if self.do_heatmap:
            batch = max(1, min(batch, int(2e9 // (width * height * 4 * 12))))     # D maps + context image per frame
            dmap_channels = 1 if self.heatmap == "raw" else 3
            # fp16 on the host like the reference (fvvdp.py:216-221); every element is written below, and page-locked
            # memory lets the batches stream back at link speed while the next batch is computed
            heatmap = self._host_buffer([1, dmap_channels, n_out, height, width])
        ctx = self._context(width, height, n_bands, planes, batch, rho_band)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        # results + the out-of-range flag share one buffer: a single device->host copy (and sync) per call
        nq = n_bands * 2 * n_out
        res = torch.zeros(nq + 2, dtype=torch.float32, device=self.device)      # Q_per_ch | range flag | JOD
        Q = res[:nq].view(n_bands, 2, n_out)
        oob = res[nq:nq + 1].view(torch.int32)

        if schedule is None:
            schedule = [min(batch, f1 - b0) for b0 in range(f0, f1, batch)]
        # pooling + JOD regression (do_pooling_and_jods, fvvdp.py:337-357) ride on the last batch
        pp = nat.PoolParams(self.beta_sch, self.beta_tch, self.beta_t, self.w_transient, self.jod_a,
                            float(10.0 ** self.log_jod_exp))
        b0 = f0
        bi = -1
        while bi + 1 < len(schedule):
            bi += 1
            nb = schedule[bi]
            idx = np.ascontiguousarray(widx[b0:b0 + fl - 1 + nb])          # history + newest frames of this batch
            try:
                feeder(ctx, idx, taps, fl, nb, oob, stream)
            except _SourceRecyclesBuffers:
                # A user source overwrote frames that the kernels of the PREVIOUS batch may still have been reading (it decodes
                # into buffers it reuses).  The feeder has switched to copying every frame on arrival; the previous batch is
                # evaluated again from intact frames (its Q columns are simply rewritten, in stream order), then this one.
                if bi > 0:
                    bi -= 1
                    b0 -= schedule[bi]
                bi -= 1
                continue
            maps_arr, dmaps = None, None
            if self.do_heatmap:                      # per-band difference maps as extra kernel outputs
                maps_arr = (nat.BandMaps * n_bands)()
                dmaps = []
                w_l, h_l = width, height
                for b in range(n_bands):
                    dmaps.append(torch.empty((nb, 2, h_l, w_l), dtype=torch.float32, device=self.device))
                    maps_arr[b].d_D = dmaps[b].data_ptr()
                    w_l, h_l = (w_l + 1) // 2, (h_l + 1) // 2
            fx, g = None, None
            if self.foveated:
                fxa = np.ascontiguousarray(fix[b0:b0 + nb], dtype=np.float32)
                if native_geometry(self.display_geometry) is not None:
                    g = C.byref(self._geom_struct())
                else:                                # user geometry: maps + gaze view directions (degrees)
                    self._set_view_maps(ctx, n_bands, width, height)
                    fxa = self._gaze_view_dirs(fxa, width, height)
                fx = nat.fptr(fxa)
            if pool and b0 + nb == f1:               # the batch that completes the clip also pools (one launch fewer)
                nat.check(nat.lib().fvvdp_bands_forward_pool(ctx.handle, nb, C.c_void_p(Q.data_ptr()), n_out, b0 - f0,
                                                             fx, g, maps_arr, C.byref(pp), C.c_void_p(res[nq + 1:].data_ptr()),
                                                             stream))
            else:
                nat.check(nat.lib().fvvdp_bands_forward(ctx.handle, nb, C.c_void_p(Q.data_ptr()), n_out, b0 - f0,
                                                        fx, g, maps_arr, stream))
            if self.do_heatmap:
                self._heatmap_batch(ctx, nb, dmaps, planes, width, height, stream, heatmap, b0 - f0)
            b0 += nb

        Q_jod = res[nq + 1] if pool else None                 # pooled with the last batch (fvvdp_bands_forward_pool)
        stats = {}
        if sync:
            res_h = self._to_host(res)                       # the one host synchronisation of the call
            if self.do_heatmap and self._copy_stream is not None:
                self._copy_stream.synchronize()              # ... plus the side stream that carries the maps
            stats['Q_per_ch'] = res_h[:nq].view(n_bands, 2, n_out).numpy()
        else:
            if self.do_heatmap:
                raise RuntimeError("sync=False is not available together with heat-map output")
            stats['Q_per_ch'] = Q
            stats['range_flag'] = oob
            stats['result_buffer'] = res          # [Q_per_ch (bands x 2 x frames) | range flag (int32 bits) | JOD]: one row per pair
        stats['rho_band'] = rho_band
        stats['frames_per_second'] = vid_source.get_frames_per_second()
        stats['width'] = width
        stats['height'] = height
        stats['N_frames'] = N_frames
        if self.do_heatmap:
            stats['heatmap'] = heatmap
        if sync and int(res_h[nq:nq + 1].view(torch.int32)[0]) != 0:
            logging.warning("Pixel outside the valid range 0-1")
        if hasattr(feeder, "release"):
            feeder.release(synced=sync)
        return (Q_jod, stats)

    def _to_host(self, res):
        """Result buffer -> host through a page-locked staging buffer kept by the metric (a pageable `.cpu()` costs ~0.1 ms more per
        call: 4K x 60, 4.25 against 4.15 ms); the caller gets its own copy.  Falls back to `.cpu()` where page-locking is refused."""
        n = res.numel()
        pin = self.__dict__.get("_res_pin")
        if pin is False:                       # page-locking was refused once: do not ask again
            return res.detach().cpu()
        if pin is None or pin.numel() < n:
            try:
                pin = torch.empty(max(n, 4096), dtype=torch.float32, pin_memory=True)
            except RuntimeError:
                pin = False
            self._res_pin = pin
        if pin is False:
            return res.detach().cpu()
        pin[:n].copy_(res.detach(), non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return pin[:n].clone()

    @staticmethod
    def _host_buffer(shape):
        nbytes = 2 * int(np.prod(shape))
        if nbytes <= (4 << 30):                   # long clips: do not page-lock tens of GB of host memory
            try:
                return torch.empty(shape, dtype=torch.float16, pin_memory=True)
            except RuntimeError:                  # page-locking refused (ulimit): plain pageable memory
                pass
        return torch.empty(shape, dtype=torch.float16)

    def _heatmap_batch(self, ctx, nb, dmaps, planes, width, height, stream, heatmap, k0):
        """Difference maps of `nb` frames -> heatmap[0, :, k0:k0+nb] (fp16, host) (fvvdp.py:469-476).  The maps are
        finished on the device in the output layout and copied back asynchronously, one contiguous run per channel."""
        ptrs = (C.c_void_p * len(dmaps))(*[d.data_ptr() for d in dmaps])
        dmap = torch.empty((nb, height, width), dtype=torch.float32, device=self.device)
        beta_jod = float(np.power(10.0, self.log_jod_exp))
        nat.check(nat.lib().fvvdp_heatmap_reconstruct(ctx.handle, nb, ptrs, float(self.w_transient), beta_jod,
                                                      abs(float(self.jod_a)), C.c_void_p(dmap.data_ptr()), stream))
        if self.heatmap == "raw":
            self._copy_back(dmap.to(torch.float16).unsqueeze(0), [heatmap[0, 0, k0:k0 + nb]])
            return
        # colouring on the device (fvvdp_heatmap_colorize): tone-mapped context frame x colour map, fp16, output layout
        from .visualize_diff_map import color_tables_host
        knots, rgb = color_tables_host(self.heatmap)
        lin01 = torch.linspace(0.0, 1.0, 1024).numpy()
        out = torch.empty((3, nb, height, width), dtype=torch.float16, device=self.device)
        nat.check(nat.lib().fvvdp_heatmap_colorize(ctx.handle, nb, C.c_void_p(dmap.data_ptr()), nat.fptr(knots), nat.fptr(rgb),
                                                   len(knots), nat.fptr(lin01), C.c_void_p(out.data_ptr()),
                                                   nb * height * width, stream))
        self._copy_back(out, [heatmap[0, ch, k0:k0 + nb] for ch in range(3)])

    def _copy_back(self, src, dst_list):
        """Device -> host copies of finished maps on a side stream, so that the next batch's kernels run meanwhile
        (page-locked destination; with pageable memory the copy is synchronous anyway)."""
        main = torch.cuda.current_stream(self.device)
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        done = torch.cuda.Event()
        done.record(main)
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(done)
            for i, dst in enumerate(dst_list):
                dst.copy_(src[i], non_blocking=True)
        src.record_stream(self._copy_stream)      # the caching allocator must not hand `src` out before the copy ran

    # ---- pooling and JOD regression (Python, as in the reference) ------------------------------------------
    def do_pooling_and_jods(self, Q_per_ch, rho_band):
        """Q_per_ch [bands, temporal channels, frames] -> JOD (0-d tensor).  Minkowski pooling over spatial bands
        (beta_sch), temporal channels (beta_tch, transient weighted by w_transient) and frames (beta_t, mean).
        A leading batch dimension [K, bands, channels, frames] pools K videos at once -> [K]."""
        Q = Q_per_ch
        d = Q.dim() - 3                        # 0, or 1 with a leading batch of videos
        if Q.shape[d + 1] == 2:
            key = (str(Q.device), Q.dtype)             # cached: a host->device upload per call costs a stream sync
            if key not in self._chan_w:
                self._chan_w[key] = torch.tensor([1.0, self.w_transient], dtype=Q.dtype, device=Q.device).view(1, 2, 1)
            Q = Q * self._chan_w[key]
        Q_sc = self.lp_norm(Q, self.beta_sch, d + 0, False)
        Q_tc = self.lp_norm(Q_sc, self.beta_tch, d + 1, False)
        Q_all = self.lp_norm(Q_tc, self.beta_t, d + 2, True)
        Q_all = Q_all.reshape(Q_all.shape[0]) if d == 1 else Q_all.squeeze()
        beta_jod = 10.0 ** self.log_jod_exp
        sign = -1 if self.jod_a < 0 else 1
        Q_jod = sign * ((abs(self.jod_a) ** (1.0 / beta_jod)) * Q_all) ** beta_jod + 10.0
        return Q_jod if d == 1 else Q_jod.squeeze()

    def lp_norm(self, x, p, dim=0, normalize=True):
        N = x.shape[dim] if normalize else 1.0
        return torch.norm(x, p, dim=dim, keepdim=True) / (float(N) ** (1. / p))

    def get_temporal_filters(self, frames_per_s):
        """Sustained (log-Gaussian) and transient (its scaled derivative) temporal filters, fp32 [2, filter_len];
        tap k weights the frame k steps in the past.  Evaluated on the host."""
        t = torch.linspace(0.0, self.filter_len / frames_per_s, self.filter_len)
        F = torch.zeros((2, t.shape[0]))
        sigma = torch.tensor([self.sustained_sigma])
        beta = torch.tensor([self.sustained_beta])
        F[0] = torch.exp(-torch.pow(torch.log(t + 1e-4) - torch.log(beta), 2.0) / (2.0 * (sigma ** 2.0)))
        F[0] = F[0] / torch.sum(F[0])
        k2 = 0.062170507756932
        Fdiff = F[0, 1:] - F[0, :-1]
        F[1] = k2 * torch.cat([Fdiff / (t[1] - t[0]), torch.tensor([0.0])], 0)
        omega = torch.tensor([0, 5])
        return F, omega

    def short_name(self):
        return "FovVideoVDP"

    def quality_unit(self):
        return "JOD"

    def get_info_string(self):
        standard_str = ', (' + self.display_name + ')' if self.display_name.startswith('standard_') else ''
        fv_mode = 'foveated' if self.foveated else 'non-foveated'
        return '"FovVideoVDP v{}, {:.4g} [pix/deg], Lpeak={:.5g}, Lblack={:.4g} [cd/m^2], {}{}"'.format(
            self.version, self.pix_per_deg, self.display_photometry.get_peak_luminance(),
            self.display_photometry.get_black_level(), fv_mode, standard_str)

    def write_features_to_json(self, stats, dest_fname):
        Q_per_ch = stats['Q_per_ch']
        fmap = {}
        for key, value in stats.items():
            if key not in ["Q_per_ch", "heatmap"]:
                fmap[key] = value.tolist() if isinstance(value, np.ndarray) else value
        for cc in range(Q_per_ch.shape[1]):
            for bb in range(Q_per_ch.shape[0]):
                fmap[f"t{cc}_b{bb}"] = Q_per_ch[bb, cc, :].tolist()
        with open(dest_fname, 'w', encoding='utf-8') as f:
            json.dump(fmap, f, ensure_ascii=False, indent=4)

    # ---- host-side preparation for the kernels --------------------------------------------------------------
    def native_params(self):
        p = nat.Params()
        p.mask_p = self.mask_p
        p.mask_q[0], p.mask_q[1] = self.mask_q_sust, self.mask_q_trans
        mk = self.__dict__.get("_mask_k")
        if mk is None or mk[0] != self.mask_c:                  # fp32 power like the reference's (fvvdp.py:585), once per value
            mk = (self.mask_c, float(torch.pow(torch.tensor(10.0), torch.tensor(self.mask_c))))
            self._mask_k = mk
        p.mask_k = mk[1]
        p.beta = self.beta
        p.sens_gain = 10.0 ** (self.sensitivity_correction / 20.0)
        p.lbkg_min, p.contrast_max, p.d_max = 0.1, 1000.0, 1e4
        return p

    def csf_tables_1d(self, rho_band, n_bands):
        """Non-foveated mode: the rho and eccentricity coordinates of the CSF query are constant per band, so the
        trilinear LUT interpolation reduces to a 1-D table over log2(L_bkg) per (band, temporal channel).  The
        rho-axis blend is done here with the same fp32 operations the per-pixel interpolation would use."""
        tab = np.zeros((n_bands, 2, nat.LUT_N), dtype=np.float32)
        for cc in range(2):
            lut = {k: torch.from_numpy(v) for k, v in self.csf_lut[cc].items()}
            for bb in range(n_bands):
                rho = torch.tensor([rho_band[bb]], dtype=torch.float32)
                rho_q = torch.log2(torch.clamp(rho, lut["rho"][0], lut["rho"][-1]))
                imin, imax, ifrc = _interpolants(rho_q, lut["rho_log"])
                v = lut["S_log"]                       # [Y, rho, ecc]; ecc = 0 sits exactly on knot 0
                t = v[:, imin[0], 0] * (1.0 - ifrc[0]) + v[:, imax[0], 0] * ifrc[0]
                tab[bb, cc] = t.numpy()
        return np.ascontiguousarray(self.csf_lut[0]["Y_log"], dtype=np.float32), tab

    def _context_key(self, W, H, n_bands, planes, batch, rho_band):
        """Everything the native context bakes in at creation, by VALUE: size, band frequencies (display resolution /
        distance), mode, model constants and the identity of the CSF tables.  (The reference re-reads all of these on
        every call, pyfvvdp/fvvdp.py:147-161,209-213,442-447.)"""
        prm = self.native_params()
        pv = (prm.mask_p, prm.mask_q[0], prm.mask_q[1], prm.mask_k, prm.beta, prm.sens_gain, prm.lbkg_min,
              prm.contrast_max, prm.d_max)
        lut_id = tuple((id(l["S_log"]), l["S_log"].shape) for l in self.csf_lut)
        return (W, H, n_bands, planes, batch, tuple(float(r) for r in rho_band), bool(self.foveated), pv, lut_id,
                str(self.device), self.timing is not None)

    def _context(self, W, H, n_bands, planes, batch, rho_band):
        key = self._context_key(W, H, n_bands, planes, batch, rho_band)
        if self._ctx is not None:
            have = self._ctx.key
            # a context made for a longer batch serves a shorter one (clips of different lengths in one folder: no new scratch,
            # no second choice of the level-0 ranges); everything else it bakes in must match by value
            if have[:4] == key[:4] and have[5:] == key[5:] and have[4] >= batch:
                return self._ctx
        self._drop_context()
        with torch.cuda.device(self.device):
            ctx = _Context(W, H, n_bands, planes, batch, rho_band, self.native_params())
            ctx.key = key
            ctx.luts = self.csf_lut           # keeps the arrays whose id() is part of the key alive
            if self.foveated:
                for cc in range(2):
                    l = self.csf_lut[cc]
                    nat.check(nat.lib().fvvdp_ctx_set_csf_3d(ctx.handle, cc, nat.fptr(l["S_log"]), nat.fptr(l["Y_log"]),
                                                             nat.fptr(l["rho_log"]), nat.fptr(l["ecc_sqrt"])))
            else:
                y_log, tab = self.csf_tables_1d(rho_band, n_bands)
                nat.check(nat.lib().fvvdp_ctx_set_csf_1d(ctx.handle, nat.fptr(y_log), nat.fptr(tab)))
            if self.timing is not None:
                nat.check(nat.lib().fvvdp_ctx_timing_enable(ctx.handle, 1))
        self._ctx = ctx
        return ctx

    def _batch_size(self, W, H, planes, n_out, fl=1):
        cap = 128
        if fl > 32:
            # 33..64 taps: the two-pass temporal path converts the batch's fl-1+batch source frames to fp32 luminance first
            # (2 streams x 4 B per pixel, context-owned) and addresses them through a 320-entry index table
            cap = max(1, min(cap, 320 - (fl - 1)))
        if self.batch_frames is not None:
            return max(1, min(int(self.batch_frames), n_out, cap if fl > 32 else n_out))
        per_frame = W * H * planes * 4 * 1.34          # all Gaussian levels of one frame
        budget = 24e9                                  # resident pyramid scratch (of 288 GB HBM3E)
        if fl > 32:
            budget -= (fl - 1) * W * H * 8.0           # luminance frames of the temporal window
            per_frame += W * H * 8.0                   # ... and of every frame of the batch
        return max(1, min(n_out, cap, int(max(budget, per_frame) // per_frame)))

    def _set_view_maps(self, ctx, n_bands, width, height):
        """User geometry model: evaluate its pix2view_direction / get_resolution_magnification once per band on the
        band's pixel grid (as the reference does per frame, fvvdp.py:424-437) and hand the maps to the kernels."""
        geom = self.display_geometry
        gstate = tuple(sorted((k, repr(v)) for k, v in vars(geom).items() if isinstance(v, (int, float, str, tuple, list, type(None)))))
        if getattr(ctx, "view_maps", None) is not None and ctx.view_geom is geom and ctx.view_gstate == gstate:
            return
        maps = []
        w_b, h_b = width, height
        for b in range(n_bands):
            xv = torch.linspace(0.5, w_b - 0.5, w_b, device=self.device)
            yv = torch.linspace(0.5, h_b - 0.5, h_b, device=self.device)
            xx, yy = torch.meshgrid(xv, yv, indexing='xy')
            vd = self.display_geometry.pix2view_direction(torch.tensor((w_b, h_b)), xx, yy)
            rm = self.display_geometry.get_resolution_magnification(vd)
            if rm.dim() == 0:
                rm = rm.expand(h_b, w_b)
            vx = vd[0].to(torch.float32).contiguous()
            vy = vd[1].to(torch.float32).contiguous()
            rm = rm.to(torch.float32).contiguous()
            maps.append((vx, vy, rm))
            nat.check(nat.lib().fvvdp_ctx_set_view_maps(ctx.handle, b, C.c_void_p(vx.data_ptr()), C.c_void_p(vy.data_ptr()),
                                                        C.c_void_p(rm.data_ptr()), float(rm.min()), float(rm.max())))
            w_b, h_b = (w_b + 1) // 2, (h_b + 1) // 2
        ctx.view_maps = maps          # keep the tensors alive as long as the context
        ctx.view_geom, ctx.view_gstate = geom, gstate     # strong reference: the object's id cannot be recycled

    def _gaze_view_dirs(self, fix_px, width, height):
        """Gaze positions [n,2] in frame pixels -> view directions in degrees through the user's geometry."""
        fx = torch.as_tensor(fix_px[:, 0] + 0.5, device=self.device)
        fy = torch.as_tensor(fix_px[:, 1] + 0.5, device=self.device)
        vd = self.display_geometry.pix2view_direction(torch.tensor((width, height)), fx, fy)
        return np.ascontiguousarray(torch.stack((vd[0], vd[1]), dim=1).to(torch.float32).cpu().numpy())

    def _geom_struct(self):
        d = native_geometry(self.display_geometry)
        if d is None:
            raise RuntimeError("internal: user geometry goes through _set_view_maps")
        g = nat.Geom()
        g.display_size_m[0], g.display_size_m[1] = d["display_size_m"]
        g.distance_m, g.ppd_centre = d["distance_m"], d["ppd_centre"]
        return g

    def _fixation(self, fixation_point, width, height, N):
        if fixation_point is None:
            fp = np.array([width // 2, height // 2], dtype=np.float32)
        elif isinstance(fixation_point, torch.Tensor):
            fp = fixation_point.detach().cpu().numpy().astype(np.float32)
        else:
            fp = np.asarray(fixation_point, dtype=np.float32)
        if fp.ndim == 1:
            fp = np.tile(fp.reshape(1, 2), (N, 1))
        if fp.shape != (N, 2):
            raise RuntimeError("fixation_point must be [x, y] or an [N_frames, 2] array")
        return np.ascontiguousarray(fp, dtype=np.float32)

    def _code_lut(self, photometry, nbits):
        """Luminance of every integer code value through the display model's own `forward` (user photometry
        subclasses work), kept on the device; cached by value, see display_model.code_value_tables."""
        return self._lut_dev.get(photometry, nbits, self.device)

    def _to_unit_float(self, a):
        a = a.to(self.device)
        if a.dtype is torch.float32:
            return a
        if a.dtype is torch.uint8:
            return a.to(torch.float32) / 255
        if a.dtype is torch.int16:
            return (a.to(torch.int32) & 0xFFFF).to(torch.float32) / 65535
        raise RuntimeError("Only uint8, uint16 and float32 is currently supported")

    def _place_sources(self, test, ref, need):
        """Source arrays [1, C, F, H, W] -> (test, ref) on the compute device plus the frame number -> position map (None:
        all frames are there in place).  Arrays that are already resident are used where they lie (no copy).  If one of
        them lives on the host, only the frames in `need` (sorted unique frame numbers) cross PCIe: with frame sharding a
        rank uploads its own frames plus the fl-1 frames of temporal halo, not the clip."""
        dev = self.device
        F = test.shape[2]
        if (test.device == dev and ref.device == dev) or len(need) >= F:
            out, remap = [], None
            for a in (test, ref):
                if a.device != dev:
                    self.last_h2d_bytes += a.numel() * a.element_size()
                out.append(a.to(dev).contiguous())
            return out[0], out[1], remap
        lo, hi = int(need[0]), int(need[-1])
        run = (hi - lo + 1 == len(need))               # one run of frames: a view; else (circular / pingpong history) a gather
        out = []
        for a in (test, ref):
            sub = a[:, :, lo:hi + 1] if run else a.index_select(2, torch.as_tensor(np.asarray(need, dtype=np.int64), device=a.device))
            if a.device != dev:
                self.last_h2d_bytes += sub.numel() * sub.element_size()
            out.append(sub.to(dev).contiguous())
        remap = np.full(F, -1, dtype=np.int32)
        remap[np.asarray(need, dtype=np.int64)] = np.arange(len(need), dtype=np.int32)
        return out[0], out[1], remap

    def _make_feeder(self, vs, width, height, need=None):
        """Returns feed(ctx, idx, taps, fl, n_out, oob, stream): fills pyramid level 0 of slots [0, n_out).
        `need`: sorted unique source frames the call will ask for (all frames when None)."""
        lib = nat.lib()
        HW = width * height
        if type(vs) is fvvdp_video_source_array or isinstance(vs, fvvdp_video_source_array):
            test = vs.test_video
            ref = vs.reference_video
            if test.shape[0] != 1:
                raise RuntimeError("Only batch size 1 is supported (as in the reference's sliding-window code)")
            if ref.dtype != test.dtype:
                # the reference unpacks each array by its own dtype (video_source.py:186-200): bring both to
                # normalised fp32 code values on the device, then the common float path applies
                test, ref = self._to_unit_float(test), self._to_unit_float(ref)
            dt = test.dtype
            if dt is torch.uint8:
                dtype, nbits = nat.FVVDP_U8, 8
            elif dt is torch.int16:
                dtype, nbits = nat.FVVDP_U16, 16
            elif dt is torch.float32:
                dtype, nbits = nat.FVVDP_F32, 0
            else:
                raise RuntimeError("Only uint8, uint16 and float32 is currently supported")
            C_ch = test.shape[1]
            if need is None:
                need = np.arange(test.shape[2])
            # uint8: 256-entry table through the model's own forward() (exact, lives in LDS).  uint16: the 65536-entry
            # table sits in global memory (6 gathers per pixel: K1 98 vs 37 us/frame at 4K), so stock display models are
            # evaluated in closed form on code/65535 like float input (<= 1e-6 relative); `exact_uint16 = True` on the
            # metric, or a user photometry class, keeps the table.
            use_closed = dtype == nat.FVVDP_F32 or (dtype == nat.FVVDP_U16 and not getattr(self, "exact_uint16", False))
            desc = native_eotf(vs.dm_photometry) if use_closed else None
            if dtype != nat.FVVDP_F32 or desc is not None:
                test_d, ref_d, remap = self._place_sources(test, ref, need)
                N = test_d.shape[2]
                e = nat.Eotf()
                if desc is None:
                    lut = self._code_lut(vs.dm_photometry, nbits)
                    e.kind, e.d_lut = nat.EOTF_LUT, lut.data_ptr()
                    e.L_min, e.L_max = self._lut_dev.table_range(lut)      # lets the pyramid pass drop clamps that cannot bind
                else:
                    e.kind = desc[0]
                    e.Y_peak = desc[1].get("Y_peak", 0.0)
                    e.Y_black = desc[1].get("Y_black", 0.0)
                    e.gamma = desc[1].get("gamma", 1.0)
                    e.L_min = desc[1].get("L_min", 0.0)
                    e.L_max = desc[1].get("L_max", 0.0)
                    # (on the source arrays, not the uploaded subset: a frame-sharded call warns like the unsharded one)
                    if e.kind == nat.EOTF_ABSOLUTE and dtype == nat.FVVDP_F32 and float(torch.maximum(test.max(), ref.max())) < 1:
                        logging.warning('Pixel values are very low. Perhaps images are not scaled in the absolute units of cd/m^2.')
                w = np.asarray(vs.color_to_luminance, dtype=np.float32)

                def feed(ctx, idx, taps, fl, n_out, oob, stream, slot0=0):
                    if remap is not None:
                        idx = np.ascontiguousarray(remap[idx])
                    nat.check(lib.fvvdp_temporal_channels(
                        ctx.handle, C.c_void_p(test_d.data_ptr()), C.c_void_p(ref_d.data_ptr()), dtype, C_ch,
                        N * HW, HW, C.byref(e), nat.fptr(w), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                        nat.fptr(taps), fl, n_out, slot0, C.c_void_p(oob.data_ptr()), stream))
                return feed
        if (isinstance(vs, fvvdp_video_source_yuv_frames) and native_eotf(vs.dm_photometry) is not None
                and not (hasattr(vs, "_resizing") and vs._resizing())
                and int(np.ceil(250.0 / (1000.0 / max(vs.get_frames_per_second(), 1e-9)))) <= 64):   # > 256 fps: generic path
            # raw planar YUV: unpacking, chroma upsampling, colour matrix and display model run in the HIP kernel
            test_d = vs.test_yuv.to(self.device).contiguous()
            ref_d = vs.reference_yuv.to(self.device).contiguous()
            desc = native_eotf(vs.dm_photometry)
            e = nat.Eotf()
            e.kind = desc[0]
            e.Y_peak = desc[1].get("Y_peak", 0.0)
            e.Y_black = desc[1].get("Y_black", 0.0)
            e.gamma = desc[1].get("gamma", 1.0)
            e.L_min = desc[1].get("L_min", 0.0)
            e.L_max = desc[1].get("L_max", 0.0)
            fmt = nat.YuvFormat()
            fmt.bit_depth = vs.bit_depth
            fmt.chroma_420 = 1 if vs.chroma_ss == "420" else 0
            for i, val in enumerate(np.asarray(vs.ycbcr2rgb, dtype=np.float32).reshape(-1)):
                fmt.ycbcr2rgb[i] = float(val)
            w = np.asarray(vs.color_to_luminance, dtype=np.float32)

            def feed_yuv(ctx, idx, taps, fl, n_out, oob, stream, slot0=0):
                if fl > 64:
                    raise RuntimeError("frame rate too high for the YUV path (temporal filter longer than 64 taps)")
                nat.check(lib.fvvdp_temporal_channels_yuv(
                    ctx.handle, C.c_void_p(test_d.data_ptr()), C.c_void_p(ref_d.data_ptr()), C.byref(fmt), vs.frame_elems,
                    C.byref(e), nat.fptr(w), idx.ctypes.data_as(C.POINTER(C.c_int32)), nat.fptr(taps), fl, n_out, slot0,
                    C.c_void_p(oob.data_ptr()), stream))
            return feed_yuv
        # generic sources (user subclasses, custom float photometry): luminance frames come from the source's own
        # get_*_frame (the user's code, run on the device); the kernels take over from the temporal filter on.
        return _PipelinedSourceFeeder(self, vs, width, height)


class _SourceRecyclesBuffers(Exception):
    """Raised by _PipelinedSourceFeeder the moment it sees a source reuse the buffer of a frame that was still in the temporal
    window (see there); _predict_on_device re-runs the batch whose input may have been overwritten."""


class _PipelinedSourceFeeder:
    """Frame supply for sources whose frames only exist behind their own `get_*_frame`.

    The reference fetches synchronously inside its frame loop (pyfvvdp/fvvdp.py:287-288).  Here every source frame is
    still fetched exactly once (on a side stream), nothing waits on the host -- while the kernels of batch b run, the
    Python code of the source already produces batch b+1 -- and nothing is copied: the temporal kernel reads the
    luminance tensors the source returned where they are (fvvdp_temporal_channels_frames; frames of the temporal window
    that were fetched for an earlier batch are simply kept alive).  Sources whose tensors the kernel cannot address
    (too far apart in memory, filters longer than 64 taps) go through one stacking copy per stream and batch instead.

    Contract for user sources: a tensor returned by `get_*_frame` is read in place for up to filter_len - 1 + batch later
    frames, and frames may be asked for again (random access).  A source that decodes into buffers it reuses (`return
    self._buf`) is noticed AFTER the fact -- a fresh frame arrives at the address of a frame still in the window, or the test and
    the reference frame of one fetch share an address.  By then the overwritten frames are gone and the kernels of the previous
    batch, still in flight on the caller's stream, may have read them half-written (the fetches run on a side stream that does
    not wait for those kernels).  So from that moment every frame is copied on arrival -- what the reference does for every
    source (pyfvvdp/fvvdp.py:289-291) --, the overwritten frames are fetched again, and _SourceRecyclesBuffers makes the caller
    evaluate the previous batch again from intact copies (batch k-2 and older had finished before the fetch: back-pressure).
    The fetches run on a side stream that first waits for the caller's current stream, so frames produced asynchronously
    on that stream just before the call (GPU-generated video, non_blocking uploads) are complete when they are read."""

    preferred_batch = 64       # largest batch when the caller did not choose (see batch_schedule)

    @staticmethod
    def batch_schedule(n_out, max_batch):
        """Batch sizes when the caller did not choose: 8, 16, 32, ... up to max_batch.  The kernels cannot start before the
        first batch has been fetched, so the first batch is small; later batches are fetched while the previous one is
        computed, and large batches are cheaper per frame (fewer launches, less of the temporal window converted twice:
        4K x60, resident source: 5.6 ms with batches of 16, 4.7 ms with one batch of 60).  A short tail joins the last batch."""
        sizes, left, b = [], n_out, 8
        while left > 0:
            nb = min(b, max_batch, left)
            if left - nb < max(4, nb // 2):
                nb = left if left <= max_batch else nb
            sizes.append(nb)
            left -= nb
            b *= 2
        return sizes

    def __init__(self, metric, vs, width, height):
        self.m, self.vs, self.W, self.H = metric, vs, width, height
        self.dev = metric.device
        res = getattr(metric, "_feeder_res", None)     # side stream + events live as long as the metric object
        if res is None or res[0] != self.dev:
            res = (self.dev, torch.cuda.Stream(device=self.dev), torch.cuda.Event(), [torch.cuda.Event(), torch.cuda.Event()])
            metric._feeder_res = res
        self.side = res[1]
        self.frames = {}                  # source frame -> (test tensor, reference tensor, their addresses); fp32 luminance on the device
        self.prev_done = None             # kernels of the previous batch have finished reading their frames
        # events are reused (creating one costs as much as ten frame fetches): one for "frames fetched", two alternating
        # ones for "batch finished" (the older one is still being polled when the newer one is recorded)
        self.ev_ready, self.ev_done = res[2], res[3]
        self.n_batches = 0
        self.retired = []                 # tensors of frames that left the window: freed once their last reader has finished
        self.live_ptrs = {}               # address -> source frame, of every tensor in self.frames (aliasing detection)
        self.copy_mode = False            # the source reuses its buffers: clone on arrival
        self.recycled = False             # set by _buffer_reuse: the batch in flight may have read overwritten frames
        self.waited_main = False
        self.eotf = nat.Eotf()
        self.eotf.kind = nat.EOTF_NONE

    def _conform(self, x):
        dev, f32 = self.dev, torch.float32
        if not (type(x) is torch.Tensor and x.dtype is f32 and x.is_cuda and x.device == dev and x.is_contiguous()):
            x = torch.as_tensor(x).to(device=dev, dtype=f32).contiguous()
        if x.numel() != self.H * self.W:
            raise RuntimeError("get_*_frame must return one luminance frame of %dx%d pixels" % (self.W, self.H))
        return x

    def _fetch(self, f):
        """One source frame of both streams (the per-frame host cost of a user source: kept to a few attribute reads)."""
        t = self._conform(self.vs.get_test_frame(f, device=self.dev))
        if self.copy_mode:
            t = t.clone()
        r = self._conform(self.vs.get_reference_frame(f, device=self.dev))
        if self.copy_mode:
            r = r.clone()
        tp, rp = t.data_ptr(), r.data_ptr()
        if not self.copy_mode:
            if tp == rp or tp in self.live_ptrs or rp in self.live_ptrs:
                return self._buffer_reuse(f, t, r)
            self.live_ptrs[tp] = f
            self.live_ptrs[rp] = f
        return (t, r, tp, rp)

    def _buffer_reuse(self, f, t, r):
        """The source handed out an address that a frame still in the temporal window occupies (or one address for the test
        and the reference frame): it decodes into buffers it reuses.  The new frame is intact right now unless its two halves
        share the buffer (then it is fetched again, copying); the held frames at the same addresses are gone (fetched again
        and copied); every other held frame is still intact and is copied before a later fetch can overwrite it."""
        tp, rp = t.data_ptr(), r.data_ptr()
        hit = sorted({self.live_ptrs[p] for p in (tp, rp) if p in self.live_ptrs})
        self.copy_mode = True
        self.live_ptrs = {}
        self.recycled = True                             # __call__ raises once the bookkeeping is consistent again
        if tp == rp:
            t = r = None                                 # the reference fetch overwrote the test frame
        else:
            t, r = t.clone(), r.clone()
        for g in list(self.frames):
            if g not in hit:
                h = self.frames[g]
                tt, rr = h[0].clone(), h[1].clone()
                self.frames[g] = (tt, rr, tt.data_ptr(), rr.data_ptr())
        for g in hit:
            self.frames[g] = self._fetch(g)               # copy mode: cloned on arrival
        if t is None:
            return self._fetch(f)
        return (t, r, t.data_ptr(), r.data_ptr())

    def __call__(self, ctx, idx, taps, fl, n_out, oob, stream):
        uniq = sorted(set(int(f) for f in idx))
        main = torch.cuda.current_stream(self.dev)
        if self.n_batches >= 2:
            # Back-pressure and buffer safety in one: the host waits for the kernels of the batch before the previous one.
            # The GPU still has the previous batch queued (nothing starves), at most ~3 batches of frames are alive however
            # fast the source is, everything retired so far has no reader left, and a source that recycles a buffer of a
            # retired frame writes it after its last reader.
            self.ev_done[self.n_batches & 1].synchronize()
            self.retired.clear()
        with torch.cuda.stream(self.side):
            if not self.waited_main:
                # frames may have been produced asynchronously on the caller's stream just before the call
                self.side.wait_stream(main)
                self.waited_main = True
            fresh = [f for f in uniq if f not in self.frames]
            for f in fresh:
                self.frames[f] = self._fetch(f)
            ready = self.ev_ready
            ready.record(self.side)
        main.wait_event(ready)
        if self.recycled:
            # The frames just overwritten may have been inputs of the previous batch, whose kernels are not known to have
            # finished: hand control back so that batch is evaluated again (every frame held now is a private copy; the
            # copies were made on the side stream, which the caller's stream has just been made to wait for).
            self.recycled = False
            if self.n_batches >= 1:
                raise _SourceRecyclesBuffers()
        pos = {f: k for k, f in enumerate(uniq)}
        ridx = np.asarray([pos[int(f)] for f in idx], dtype=np.int32)
        held = [self.frames[f] for f in uniq]
        tp = (C.c_void_p * len(uniq))(*[h[2] for h in held])
        rp = (C.c_void_p * len(uniq))(*[h[3] for h in held])
        lib = nat.lib()
        rc = lib.fvvdp_temporal_channels_frames(
            ctx.handle, tp, rp, len(uniq), nat.FVVDP_F32, 1, 0, C.byref(self.eotf), None,
            ridx.ctypes.data_as(C.POINTER(C.c_int32)), nat.fptr(taps), fl, n_out, 0, C.c_void_p(oob.data_ptr()), stream)
        if rc == nat.FVVDP_EUNSUPPORTED:          # one array per stream, then the general entry point
            with torch.cuda.stream(self.side):
                bt = torch.stack([h[0].reshape(-1) for h in held])
                br = torch.stack([h[1].reshape(-1) for h in held])
                ready.record(self.side)
            main.wait_event(ready)
            for b in (bt, br):
                b.record_stream(main)
            nat.check(lib.fvvdp_temporal_channels(
                ctx.handle, C.c_void_p(bt.data_ptr()), C.c_void_p(br.data_ptr()), nat.FVVDP_F32, 1, 0, self.H * self.W,
                C.byref(self.eotf), None, ridx.ctypes.data_as(C.POINTER(C.c_int32)), nat.fptr(taps), fl, n_out, 0,
                C.c_void_p(oob.data_ptr()), stream))
        else:
            nat.check(rc)
        # The frames were allocated on the side stream and are read by kernels torch does not know about: they are kept
        # alive here until an event recorded AFTER their last reader has passed (cheaper than record_stream on every
        # tensor and batch: 184 calls on a 60-frame clip); release() covers the tensors still held when the call ends.
        keep = set(uniq)
        for f in [f for f in self.frames if f not in keep]:
            h = self.frames.pop(f)
            self.live_ptrs.pop(h[2], None)
            self.live_ptrs.pop(h[3], None)
            self.retired.append(h)                  # not in this batch's window: last read by the previous batch or earlier
        done = self.ev_done[self.n_batches & 1]
        self.n_batches += 1
        done.record(main)
        self.prev_done = done

    def release(self, synced):
        """End of the predict call.  After a host synchronisation nothing is in flight; otherwise (sync=False) the
        allocator is told that the main stream still reads the frames."""
        if not synced:
            main = torch.cuda.current_stream(self.dev)
            for held in list(self.frames.values()) + self.retired:
                held[0].record_stream(main)
                held[1].record_stream(main)
        self.frames.clear()
        self.retired.clear()
        self.live_ptrs = {}

