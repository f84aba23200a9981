"""Frame supply protocol (same names and contracts as pyfvvdp/video_source.py of the reference).

`fvvdp_video_source` is the abstract protocol user code may implement; `fvvdp_video_source_array` wraps in-memory
tensors/arrays.  For the array source the metric does NOT call `get_*_frame`: the HIP path reads the packed source
arrays directly (unpack + photometry + luminance are fused with the temporal filter on the GPU).  `get_*_frame` is
kept for API compatibility and for callers that want single luminance frames.
"""
from abc import abstractmethod

import numpy as np
import torch
from torch.functional import Tensor

from . import utils
from .display_model import fvvdp_display_photometry


class fvvdp_video_source:
    @abstractmethod
    def get_video_size(self):
        """(height, width, frames)"""

    @abstractmethod
    def get_frames_per_second(self) -> int:
        pass

    @abstractmethod
    def get_test_frame(self, frame, device) -> Tensor:
        """fp32 luminance [1,1,1,H,W] in cd/m^2"""

    @abstractmethod
    def get_reference_frame(self, frame, device) -> Tensor:
        pass


def reshuffle_dims(T: Tensor, in_dims: str, out_dims: str) -> Tensor:
    """Permute `T` from dimension order `in_dims` (e.g. "HWC") to `out_dims` (e.g. "BCFHW"); dimensions missing
    from `in_dims` become singletons."""
    in_dims, out_dims = in_dims.upper(), out_dims.upper()
    if in_dims == out_dims:
        return T                      # already in the requested order (the common call: "BCFHW" tensors)
    present = [d for d in out_dims if d in in_dims]
    T_p = T.permute([in_dims.index(d) for d in present])
    shape = [T_p.shape[present.index(d)] if d in present else 1 for d in out_dims]
    return T_p.reshape(shape)


class fvvdp_video_source_dm(fvvdp_video_source):
    """Video source that turns display-encoded content into luminance with a photometric display model."""

    def __init__(self, display_photometry='sdr_4k_30', color_space_name='sRGB'):
        colorspaces = utils.config_files.load("color_spaces.json")
        if color_space_name not in colorspaces:
            raise RuntimeError("Unknown color space: \"" + color_space_name + "\"")
        self.color_to_luminance = colorspaces[color_space_name]['RGB2Y']
        if isinstance(display_photometry, str):
            self.dm_photometry = fvvdp_display_photometry.load(display_photometry)
        elif isinstance(display_photometry, fvvdp_display_photometry):
            self.dm_photometry = display_photometry
        else:
            raise RuntimeError("display_model must be a string or fvvdp_display_photometry subclass")


class fvvdp_video_source_array(fvvdp_video_source_dm):
    """Test/reference content held in torch tensors or numpy arrays (uint8, uint16, float32), any dimension order."""

    def __init__(self, test_video, reference_video, fps, dim_order='BCFHW', display_photometry='sdr_4k_30',
                 color_space_name='sRGB'):
        super().__init__(display_photometry=display_photometry, color_space_name=color_space_name)
        if test_video.shape != reference_video.shape:
            raise RuntimeError('Test and reference image/video tensors must be exactly the same shape')
        if len(dim_order) != len(test_video.shape):
            raise RuntimeError('Input tensor much have exactly as many dimensions as there are characters in the "dims" parameter')
        test_video = self._as_tensor(test_video)
        reference_video = self._as_tensor(reference_video)
        test_video = reshuffle_dims(test_video, in_dims=dim_order, out_dims="BCFHW")
        reference_video = reshuffle_dims(reference_video, in_dims=dim_order, out_dims="BCFHW")
        B, C, F, H, W = test_video.shape
        if fps == 0 and F > 1:
            raise RuntimeError('When passing video sequences, you must set ''frames_per_second'' parameter')
        if C != 3 and C != 1:
            raise RuntimeError('The content must have either 1 or 3 colour channels.')
        self.fps = fps
        self.is_video = (fps > 0)
        self.is_color = (C == 3)
        self.test_video = test_video
        self.reference_video = reference_video

    @staticmethod
    def _as_tensor(v):
        if isinstance(v, np.ndarray):
            if v.dtype == np.uint16:
                v = v.view(np.int16)        # torch has no uint16 arithmetic: carry the bits in int16
            v = torch.from_numpy(np.ascontiguousarray(v))
        return v

    def get_frames_per_second(self):
        return self.fps

    def get_video_size(self):
        sh = self.test_video.shape
        return (sh[3], sh[4], sh[2])

    def get_test_frame(self, frame, device=torch.device('cpu')):
        return self._get_frame(self.test_video, frame, device)

    def get_reference_frame(self, frame, device=torch.device('cpu')):
        return self._get_frame(self.reference_video, frame, device)

    def _get_frame(self, from_array, frame, device):
        fr = from_array[:, :, frame:(frame + 1), :, :].to(device)
        if from_array.dtype is torch.float32:
            V = fr
        elif from_array.dtype is torch.int16:
            V = (fr.to(torch.int32) & 0xFFFF).to(torch.float32) / 65535
        elif from_array.dtype is torch.uint8:
            V = fr.to(torch.float32) / 255
        else:
            raise RuntimeError("Only uint8, uint16 and float32 is currently supported")
        L = self.dm_photometry.forward(V)
        if self.is_color:
            w = self.color_to_luminance
            L = L[:, 0:1] * w[0] + L[:, 1:2] * w[1] + L[:, 2:3] * w[2]
        return L
