"""Raw planar YUV frames as a video source.

This is the data the reference's file reader hands to its `video_reader_yuv_pytorch.unpack`
(pyfvvdp/video_source_file.py:166-276): per frame the Y plane, then U, then V, limited range, 8 bit (uint8) or
10-16 bit (uint16).  With this source class the metric feeds the raw planes straight to the GPU: fixed->float,
4:2:0 chroma upsampling, YCbCr->RGB, display model, luminance and the temporal filter run in one HIP kernel
(`fvvdp_temporal_channels_yuv`).  `get_*_frame` implement the same arithmetic with torch ops for callers that want
single luminance frames; the metric itself does not use them for this class -- unless `full_screen_resize` is set (the
CLI's --full-screen-resize, run_fvvdp.py:84: both videos are resized in RGB to the display's resolution,
video_source_file.py:238-244).  Then every frame is unpacked, resized (torch's `interpolate` arithmetic: bilinear, bicubic,
nearest or area), clipped, passed through the display model and turned into luminance by two HIP kernels
(`fvvdp_yuv_frame_resized`), and the metric takes these luminance frames through its per-frame feeder.
"""
import ctypes as C

import numpy as np
import torch

from .video_source import fvvdp_video_source_dm

YCBCR2RGB = {
    "bt2020nc": [[1, 0, 1.47460], [1, -0.16455, -0.57135], [1, 1.88140, 0]],
    "bt709": [[1, 0, 1.402], [1, -0.344136, -0.714136], [1, 1.772, 0]],
}


class fvvdp_video_source_yuv_frames(fvvdp_video_source_dm):
    def __init__(self, test_yuv, reference_yuv, fps, width, height, bit_depth=8, chroma_ss="420", color_space="bt709",
                 display_photometry='sdr_4k_30', color_space_name='auto', full_screen_resize=None, resize_resolution=None):
        if color_space_name == 'auto':
            color_space_name = "BT.2020" if color_space == 'bt2020nc' else "sRGB"
        super().__init__(display_photometry=display_photometry, color_space_name=color_space_name)
        if chroma_ss not in ("420", "444"):
            raise RuntimeError(f"Unrecognized chroma subsampling {chroma_ss}")
        if chroma_ss == "420" and (width % 2 or height % 2):
            raise RuntimeError("4:2:0 video needs even width and height")
        if not (8 <= bit_depth <= 16):
            raise RuntimeError("bit depth must be 8..16")
        self.width, self.height, self.fps = int(width), int(height), fps
        self.bit_depth, self.chroma_ss, self.color_space = int(bit_depth), chroma_ss, color_space
        self.y_pixels = self.width * self.height
        self.uv_shape = (self.height // 2, self.width // 2) if chroma_ss == "420" else (self.height, self.width)
        self.uv_pixels = self.uv_shape[0] * self.uv_shape[1]
        self.frame_elems = self.y_pixels + 2 * self.uv_pixels
        self.test_yuv = self._as_frames(test_yuv)
        self.reference_yuv = self._as_frames(reference_yuv)
        if self.test_yuv.shape != self.reference_yuv.shape:
            raise RuntimeError('Test and reference image/video tensors must be exactly the same shape')
        self.frames = self.test_yuv.shape[0]
        self.ycbcr2rgb = YCBCR2RGB["bt2020nc" if color_space == "bt2020nc" else "bt709"]
        if full_screen_resize is not None:
            if full_screen_resize not in ("bilinear", "bicubic", "nearest", "area"):          # run_fvvdp.py:84
                raise RuntimeError(f"Unknown resize method '{full_screen_resize}'")
            if resize_resolution is None:
                raise RuntimeError("full_screen_resize needs resize_resolution=(width, height)")
        self.full_screen_resize = full_screen_resize
        self.resize_resolution = None if resize_resolution is None else (int(resize_resolution[0]), int(resize_resolution[1]))
        self._resize_ws = {}

    def _resizing(self):
        """video_source_file.py:238-239: resize only when a method is given AND the size differs."""
        return self.full_screen_resize is not None and self.resize_resolution != (self.width, self.height)

    def _as_frames(self, a):
        if isinstance(a, np.ndarray):
            if a.dtype == np.uint16:
                a = a.view(np.int16)
            a = torch.from_numpy(np.ascontiguousarray(a))
        if a.dtype is torch.int32:                       # exact 10-16 bit values: carry as int16 bit patterns
            a = a.to(torch.int16)
        want = torch.uint8 if self.bit_depth == 8 else torch.int16
        if a.dtype is not want:
            raise RuntimeError("YUV frames must be uint8 (8 bit) or uint16 (10-16 bit)")
        a = a.reshape(-1, self.frame_elems) if a.numel() % self.frame_elems == 0 else None
        if a is None:
            raise RuntimeError("YUV buffer size is not a multiple of the frame size")
        return a.contiguous()

    def get_video_size(self):
        if self._resizing():                              # video_source_file.py:326-327
            return (self.resize_resolution[1], self.resize_resolution[0], self.frames)
        return (self.height, self.width, self.frames)

    def get_frames_per_second(self):
        return self.fps

    def get_test_frame(self, frame, device=torch.device('cpu')):
        return self._get_frame(self.test_yuv, frame, device)

    def get_reference_frame(self, frame, device=torch.device('cpu')):
        return self._get_frame(self.reference_yuv, frame, device)

    def unpack(self, frame, device):
        """One raw frame -> display-encoded RGB [H,W,3] in [0,1] (torch ops; with `full_screen_resize`: resized before the clip,
        as video_source_file.py:238-244)."""
        x = frame.to(device)
        x = (x.to(torch.int32) & 0xFFFF).to(torch.float32) if x.dtype is torch.int16 else x.to(torch.float32)
        sc = 2 ** (self.bit_depth - 8)
        Y = torch.clip(x[:self.y_pixels] * (1 / (sc * 219)) - 16 / 219, 0, 1).reshape(self.height, self.width)
        uv = torch.clip(x[self.y_pixels:] * (1 / (sc * 224)) - 128 / 224, -0.5, 0.5).reshape(1, 2, *self.uv_shape)
        if self.chroma_ss == "420":
            uv = torch.nn.functional.interpolate(uv, scale_factor=2, mode='bilinear')
        Yuv = torch.cat((Y[None], uv[0]), 0).permute(1, 2, 0)
        M = torch.tensor(self.ycbcr2rgb, dtype=torch.float32, device=device)
        RGB = Yuv @ M.transpose(1, 0)
        if self._resizing():
            RGB = torch.nn.functional.interpolate(RGB.permute(2, 0, 1)[None], size=(self.resize_resolution[1], self.resize_resolution[0]),
                                                  mode=self.full_screen_resize)[0].permute(1, 2, 0)
        return RGB.clip(0, 1)

    def _get_frame(self, frames, frame, device):
        h, w_, _ = self.get_video_size()
        device = torch.device(device)
        if self._resizing() and device.type == "cuda":
            from .display_model import native_eotf
            desc = native_eotf(self.dm_photometry)
            if desc is not None:
                return self._get_frame_native(frames, frame, device, desc)
        rgb = self.unpack(frames[frame], device).permute(2, 0, 1).reshape(1, 3, 1, h, w_)
        L = self.dm_photometry.forward(rgb)
        w = self.color_to_luminance
        return L[:, 0:1] * w[0] + L[:, 1:2] * w[1] + L[:, 2:3] * w[2]

    def _get_frame_native(self, frames, frame, device, desc, want_rgb=False):
        """Resized luminance frame [1,1,1,H',W'] by the HIP kernels (`fvvdp_yuv_frame_resized`); `want_rgb`: also the clipped RGB
        [3,H',W'] that `unpack` returns (parity tests)."""
        from . import _native as nat
        wo, ho = self.resize_resolution
        with torch.cuda.device(device):
            x = frames[frame].to(device, non_blocking=True).contiguous()
            # workspace = the unclipped RGB planes at the source resolution, reused by every frame fetched on the same stream (stream order
            # keeps two frames apart; callers that fetch on several streams get one workspace per stream)
            wkey = (device, torch.cuda.current_stream(device).cuda_stream)
            ws = self._resize_ws.get(wkey)
            if ws is None:
                if len(self._resize_ws) >= 4:
                    self._resize_ws.clear()
                ws = torch.empty(3 * self.height * self.width, dtype=torch.float32, device=device)
                self._resize_ws[wkey] = ws
            lum = torch.empty((1, 1, 1, ho, wo), dtype=torch.float32, device=device)
            rgb = torch.empty((3, ho, wo), dtype=torch.float32, device=device) if want_rgb else None
            fmt = nat.YuvFormat()
            fmt.bit_depth, fmt.chroma_420 = self.bit_depth, 1 if self.chroma_ss == "420" else 0
            for i, v in enumerate(np.asarray(self.ycbcr2rgb, dtype=np.float32).reshape(-1)):
                fmt.ycbcr2rgb[i] = float(v)
            e = nat.Eotf()
            e.kind = desc[0]
            e.Y_peak, e.Y_black = desc[1].get("Y_peak", 0.0), desc[1].get("Y_black", 0.0)
            e.gamma = desc[1].get("gamma", 1.0)
            e.L_min, e.L_max = desc[1].get("L_min", 0.0), desc[1].get("L_max", 0.0)
            w = np.ascontiguousarray(np.asarray(self.color_to_luminance, dtype=np.float32))
            nat.check(nat.lib().fvvdp_yuv_frame_resized(
                C.c_void_p(x.data_ptr()), C.byref(fmt), self.width, self.height, C.c_void_p(ws.data_ptr()), wo, ho,
                nat.RESIZE_MODES[self.full_screen_resize], C.byref(e), nat.fptr(w), C.c_void_p(lum.data_ptr()),
                C.c_void_p(rgb.data_ptr()) if want_rgb else None, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
            x.record_stream(torch.cuda.current_stream(device))
        return (lum, rgb) if want_rgb else lum
