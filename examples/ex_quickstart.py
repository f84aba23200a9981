#!/usr/bin/env python3
"""Tour of the API on synthetic data (runs on one MI355X in a few seconds).

    python examples/ex_quickstart.py

Every call below has the signature of the corresponding `pyfvvdp` call; `import fovvideovdp_amd as pyfvvdp` is the only
change an existing script needs for the parts of the reference covered here.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fovvideovdp_amd as pyfvvdp
from fovvideovdp_amd.synth import synth_video_pair, synth_yuv_pair

# --- video, uint8 RGB, tensors already on the GPU (BCFHW like the reference) ---------------------------------------
test, ref = synth_video_pair(30, 1080, 1920, device="cuda")
fv = pyfvvdp.fvvdp(display_name="standard_fhd")
q, stats = fv.predict(test, ref, dim_order="BCFHW", frames_per_second=30)
print("video          : %.4f JOD, Q_per_ch %s" % (float(q), stats["Q_per_ch"].shape))

# --- still image from host memory, with a difference map ------------------------------------------------------------
img_t = test[0, :, 0].permute(1, 2, 0).cpu().numpy()            # HWC uint8
img_r = ref[0, :, 0].permute(1, 2, 0).cpu().numpy()
fvh = pyfvvdp.fvvdp(display_name="standard_fhd", heatmap="threshold")
q, stats = fvh.predict(img_t, img_r, dim_order="HWC")
print("image + heatmap: %.4f JOD, heatmap %s %s" % (float(q), tuple(stats["heatmap"].shape), stats["heatmap"].dtype))

# --- foveated, gaze moving across the frame, HDR display --------------------------------------------------------------
gaze = np.stack([np.linspace(200, 1700, 30), np.full(30, 540.0)], 1)
fvf = pyfvvdp.fvvdp(display_name="standard_hdr_pq", foveated=True)
q, _ = fvf.predict(test, ref, frames_per_second=30, fixation_point=gaze)
print("foveated PQ    : %.4f JOD" % float(q))

# --- raw planar YUV 4:2:0 frames (what a decoder's rawvideo pipe or a .yuv file holds) ------------------------------------
ty, ry = synth_yuv_pair(30, 1080, 1920, bit_depth=8, chroma_ss="420", device="cuda")
vs = pyfvvdp.fvvdp_video_source_yuv_frames(ty, ry, 30, 1920, 1080, bit_depth=8, chroma_ss="420", color_space="bt709",
                                          display_photometry=fv.display_photometry)
q, _ = fv.predict_video_source(vs)
print("YUV 4:2:0      : %.4f JOD" % float(q))

# --- PU21-PSNR side metric ---------------------------------------------------------------------------------------------------
psnr, _ = pyfvvdp.pu_psnr(display_name="standard_fhd").predict(test, ref, frames_per_second=30)
print("PU21-PSNR      : %.3f dB" % float(psnr))
