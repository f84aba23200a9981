"""MI355X-native FovVideoVDP hot path behind the reference's Python API."""
from .fvvdp import fvvdp
from .pupsnr import pu_psnr
from .display_model import (fvvdp_display_photometry, fvvdp_display_photo_eotf, fvvdp_display_photo_gog,
                            fvvdp_display_photo_absolute, fvvdp_display_geometry)
from .video_source import fvvdp_video_source, fvvdp_video_source_dm, fvvdp_video_source_array, reshuffle_dims
from .video_source_yuv import fvvdp_video_source_yuv_frames

__all__ = ["fvvdp", "pu_psnr", "fvvdp_display_photometry", "fvvdp_display_photo_eotf", "fvvdp_display_photo_gog",
           "fvvdp_display_photo_absolute", "fvvdp_display_geometry", "fvvdp_video_source",
           "fvvdp_video_source_dm", "fvvdp_video_source_array", "fvvdp_video_source_yuv_frames", "reshuffle_dims"]
