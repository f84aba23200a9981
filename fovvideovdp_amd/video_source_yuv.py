"""Raw planar YUV frames as a video source.

This is the data the reference's file reader hands to its `video_reader_yuv_pytorch.unpack`
(pyfvvdp/video_source_file.py:166-276): per frame the Y plane, then U, then V, limited range, 8 bit (uint8) or
10-16 bit (uint16).  With this source class the metric feeds the raw planes straight to the GPU: fixed->float,
4:2:0 chroma upsampling, YCbCr->RGB, display model, luminance and the temporal filter run in one HIP kernel
(`fvvdp_temporal_channels_yuv`).  `get_*_frame` implement the same arithmetic with torch ops for callers that want
single luminance frames; the metric itself does not use them for this class.
"""
import numpy as np
import torch

from .video_source import fvvdp_video_source_dm

YCBCR2RGB = {
    "bt2020nc": [[1, 0, 1.47460], [1, -0.16455, -0.57135], [1, 1.88140, 0]],
    "bt709": [[1, 0, 1.402], [1, -0.344136, -0.714136], [1, 1.772, 0]],
}


class fvvdp_video_source_yuv_frames(fvvdp_video_source_dm):
    def __init__(self, test_yuv, reference_yuv, fps, width, height, bit_depth=8, chroma_ss="420", color_space="bt709",
                 display_photometry='sdr_4k_30', color_space_name='auto'):
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
        return (self.height, self.width, self.frames)

    def get_frames_per_second(self):
        return self.fps

    def get_test_frame(self, frame, device=torch.device('cpu')):
        return self._get_frame(self.test_yuv, frame, device)

    def get_reference_frame(self, frame, device=torch.device('cpu')):
        return self._get_frame(self.reference_yuv, frame, device)

    def unpack(self, frame, device):
        """One raw frame -> display-encoded RGB [H,W,3] in [0,1]."""
        x = frame.to(device)
        x = (x.to(torch.int32) & 0xFFFF).to(torch.float32) if x.dtype is torch.int16 else x.to(torch.float32)
        sc = 2 ** (self.bit_depth - 8)
        Y = torch.clip(x[:self.y_pixels] * (1 / (sc * 219)) - 16 / 219, 0, 1).reshape(self.height, self.width)
        uv = torch.clip(x[self.y_pixels:] * (1 / (sc * 224)) - 128 / 224, -0.5, 0.5).reshape(1, 2, *self.uv_shape)
        if self.chroma_ss == "420":
            uv = torch.nn.functional.interpolate(uv, scale_factor=2, mode='bilinear')
        Yuv = torch.cat((Y[None], uv[0]), 0).permute(1, 2, 0)
        M = torch.tensor(self.ycbcr2rgb, dtype=torch.float32, device=device)
        return (Yuv @ M.transpose(1, 0)).clip(0, 1)

    def _get_frame(self, frames, frame, device):
        rgb = self.unpack(frames[frame], device).permute(2, 0, 1).reshape(1, 3, 1, self.height, self.width)
        L = self.dm_photometry.forward(rgb)
        w = self.color_to_luminance
        return L[:, 0:1] * w[0] + L[:, 1:2] * w[1] + L[:, 2:3] * w[2]
