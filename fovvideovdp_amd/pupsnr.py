"""PU21-PSNR side metric on the MI355X ingest path.

Mirror of the reference's `pyfvvdp.pu_psnr` (pyfvvdp/pupsnr.py): frames are converted to luminance by the video
source's display model, encoded with the perceptually uniform PU21 transform and compared with PSNR, averaged over
the frames.  Array sources run in one fused HIP kernel (`fvvdp_pu21_sse`); other sources hand over the luminance
frames their own `get_*_frame` produce.

Difference from the reference: its `predict()` reads `self.display_photometry` / `self.color_space`, which its
constructor never sets (pupsnr.py:45), so only `predict_video_source` can run there.  Here the constructor accepts
the display description (`display_name`, `display_photometry`, `color_space`; defaults as in `fvvdp`) and
`predict()` works.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as nat
from .display_model import code_value_tables, fvvdp_display_photometry, native_eotf
from .video_source import fvvdp_video_source_array


class PU:
    """PU21 encoding parameters (pyfvvdp/utils.py:157-181); `encode` itself runs in the kernel."""
    _TYPES = {
        'banding': [1.063020987, 0.4200327408, 0.1666005322, 0.2817030548, 1.029472678, 1.119265011, 502.1303377],
        'banding_glare': [234.0235618, 216.9339286, 0.0001091864237, 0.893206924, 0.06733984121, 1.444718567, 567.6315065],
        'peaks': [1.057454135, 0.6234292574, 0.3060331179, 0.3702234502, 1.116868695, 1.109926637, 391.3707005],
        'peaks_glare': [1.374063733, 0.3160810744, 0.1350497609, 0.510558148, 1.049265455, 1.404963498, 427.3579761],
    }

    def __init__(self, L_min=0.005, L_max=10000, type='banding_glare'):
        if type not in self._TYPES:
            raise ValueError(f'Unknown type: {type}')
        self.L_min, self.L_max = L_min, L_max
        self.p = self._TYPES[type]
        p = self.p
        self.peak = p[6] * (((p[0] + p[1] * L_max ** p[3]) / (1 + p[2] * L_max ** p[3])) ** p[4] - p[5])

    def native(self):
        s = nat.Pu21()
        for i in range(7):
            s.p[i] = self.p[i]
        s.L_min, s.L_max = self.L_min, self.L_max
        return s


class pu_psnr:
    def __init__(self, device=None, display_name="standard_4k", display_photometry=None, color_space="sRGB"):
        if device is None:
            device = torch.device('cuda:0')
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("fovvideovdp_amd runs on an AMD GPU only (no CPU fallback)")
        nat.lib()                                  # fail loudly if the HIP library is missing
        self.pu = PU()
        self.display_photometry = (fvvdp_display_photometry.load(display_name) if display_photometry is None
                                   else display_photometry)
        self.color_space = color_space
        self._lut_dev = code_value_tables()

    def predict(self, test_cont, reference_cont, dim_order="BCFHW", frames_per_second=0, fixation_point=None,
                frame_padding="replicate"):
        vs = fvvdp_video_source_array(test_cont, reference_cont, frames_per_second, dim_order=dim_order,
                                      display_photometry=self.display_photometry, color_space_name=self.color_space)
        return self.predict_video_source(vs, fixation_point=fixation_point, frame_padding=frame_padding)

    def predict_video_source(self, vid_source, fixation_point=None, frame_padding="replicate"):
        """Returns (PU21-PSNR in dB as a 0-d tensor, None) like the reference (pupsnr.py:52-76)."""
        height, width, N = vid_source.get_video_size()
        HW = height * width
        sse = torch.empty(N, dtype=torch.float64, device=self.device)
        oob = torch.zeros(1, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if isinstance(vid_source, fvvdp_video_source_array) and self._array_sse(vid_source, HW, N, sse, oob, stream):
                pass
            else:
                self._generic_sse(vid_source, HW, N, sse, oob, stream)
        mse = (sse / float(HW)).to(torch.float32)                 # per frame, fp32 from here on like torch.mean
        psnr = (20.0 * torch.log10(float(self.pu.peak) / torch.sqrt(mse))).to(torch.float64).mean()
        return psnr.to(torch.float32), None

    # ---- host side of fvvdp_pu21_sse ----------------------------------------------------------------------
    def _code_lut(self, photometry, nbits):
        return self._lut_dev.get(photometry, nbits, self.device)

    def _call(self, test_d, ref_d, dtype, C_ch, chan_stride, frame_stride, HW, e, w, n, sse, oob, stream):
        partial = torch.empty(n * nat.PSNR_SLICES, dtype=torch.float64, device=self.device)
        pu = self.pu.native()
        nat.check(nat.lib().fvvdp_pu21_sse(C.c_void_p(test_d.data_ptr()), C.c_void_p(ref_d.data_ptr()), dtype, C_ch,
                                           chan_stride, frame_stride, HW, C.byref(e), nat.fptr(w) if w is not None else None,
                                           C.byref(pu), n, C.c_void_p(partial.data_ptr()), C.c_void_p(sse.data_ptr()),
                                           C.c_void_p(oob.data_ptr()), stream))

    def _array_sse(self, vs, HW, N, sse, oob, stream):
        test, ref = vs.test_video, vs.reference_video
        if test.shape[0] != 1:
            raise RuntimeError("Only batch size 1 is supported")
        if ref.dtype != test.dtype:              # each array is unpacked by its own dtype (video_source.py:186-200)
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
        e = nat.Eotf()
        if dtype != nat.FVVDP_F32:
            lut = self._code_lut(vs.dm_photometry, nbits)
            e.kind, e.d_lut = nat.EOTF_LUT, lut.data_ptr()
        else:
            desc = native_eotf(vs.dm_photometry)
            if desc is None:
                return False                     # custom photometry on float data: the source's own forward()
            e.kind = desc[0]
            e.Y_peak = desc[1].get("Y_peak", 0.0)
            e.Y_black = desc[1].get("Y_black", 0.0)
            e.gamma = desc[1].get("gamma", 1.0)
            e.L_min = desc[1].get("L_min", 0.0)
            e.L_max = desc[1].get("L_max", 0.0)
        C_ch = test.shape[1]
        test_d = test.to(self.device).contiguous()           # [1, C, N, H, W]
        ref_d = ref.to(self.device).contiguous()
        w = np.asarray(vs.color_to_luminance, dtype=np.float32) if C_ch == 3 else None
        self._call(test_d, ref_d, dtype, C_ch, N * HW, HW, HW, e, w, N, sse, oob, stream)
        return True

    def _to_unit_float(self, a):
        a = a.to(self.device)
        if a.dtype is torch.float32:
            return a
        if a.dtype is torch.uint8:
            return a.to(torch.float32) / 255
        if a.dtype is torch.int16:
            return (a.to(torch.int32) & 0xFFFF).to(torch.float32) / 65535
        raise RuntimeError("Only uint8, uint16 and float32 is currently supported")

    def _generic_sse(self, vs, HW, N, sse, oob, stream):
        e = nat.Eotf()
        e.kind = nat.EOTF_NONE
        step = 16
        for f0 in range(0, N, step):
            n = min(step, N - f0)
            T = torch.stack([vs.get_test_frame(f, device=self.device).reshape(-1) for f in range(f0, f0 + n)]).to(torch.float32).contiguous()
            R = torch.stack([vs.get_reference_frame(f, device=self.device).reshape(-1) for f in range(f0, f0 + n)]).to(torch.float32).contiguous()
            self._call(T, R, nat.FVVDP_F32, 1, HW, HW, HW, e, None, n, sse[f0:f0 + n], oob, stream)

    def short_name(self):
        return "PU21-PSNR"

    def quality_unit(self):
        return "dB"

    def get_info_string(self):
        return None
