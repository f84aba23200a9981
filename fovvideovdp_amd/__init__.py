"""MI355X-native FovVideoVDP hot path behind the reference's Python API."""
from .fvvdp import fvvdp
from .pupsnr import pu_psnr
from .display_model import (fvvdp_display_photometry, fvvdp_display_photo_eotf, fvvdp_display_photo_gog,
                            fvvdp_display_photo_absolute, fvvdp_display_geometry)
from .video_source import fvvdp_video_source, fvvdp_video_source_dm, fvvdp_video_source_array, reshuffle_dims
from .video_source_yuv import fvvdp_video_source_yuv_frames

__all__ = ["fvvdp", "pu_psnr", "fvvdp_display_photometry", "fvvdp_display_photo_eotf", "fvvdp_display_photo_gog",
           "fvvdp_display_photo_absolute", "fvvdp_display_geometry", "fvvdp_video_source",
           "fvvdp_video_source_dm", "fvvdp_video_source_array", "fvvdp_video_source_yuv_frames", "reshuffle_dims", "load_image_as_array"]


def load_image_as_array(imgfile):
    """Exported by the reference's package (pyfvvdp/__init__.py:3, video_source_file.py:29).  File decoding is outside the path
    this package accelerates; the reader lives with the other file readers in examples/file_sources.py of the source tree and is
    imported from there on first use."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "file_sources.py")
    if not os.path.isfile(path):
        raise RuntimeError("load_image_as_array needs examples/file_sources.py of the source tree (file readers are not part of "
                           "the installed package); read the image with any library and pass the array to fvvdp.predict")
    spec = importlib.util.spec_from_file_location("fovvideovdp_amd_file_sources", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.load_image_as_array(imgfile)
