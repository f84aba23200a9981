// Launchers of the stage-1 (temporal) kernels.  The kernels are instantiated in temporal_launch.hip, which is compiled once
// per sample type (-DK1_PART=0 uint8, 1 uint16, 2 float, 3 planar YUV): the register-ring kernels are fully unrolled per ring
// length, channel count and display model, and one translation unit holding all of them took 4 minutes to compile.
#pragma once
#include "temporal_kernels.hpp"

// Pixels per lane (PX) and frames of raw samples in flight (TD) of temporal_vec_kernel, per ring length and sample type.
// Measured at 4K (tools/gpu_fps.py): the long rings are register-bound (2*FL*PX ring registers), float samples are 4x
// wider than 8-bit ones in the prefetch registers.
#ifndef K1_PX8
#define K1_PX8 4
#endif
#ifndef K1_PXF8
#define K1_PXF8 4         // float samples, 8-slot ring
#endif
#ifndef K1_PX16
#define K1_PX16 2
#endif
#ifndef K1_PX32
#define K1_PX32 2
#endif
#ifndef K1_TD8
#define K1_TD8 1
#endif
#ifndef K1_TD16
#define K1_TD16 1
#endif
#ifndef K1_TD32
#define K1_TD32 1
#endif
// what the 64-slot ring (temporal filters of 33-64 taps, 129-256 fps) is instantiated for
static inline bool k1_ring64_ok(int dtype, int C, int eotf_kind) {
    if (dtype == FVVDP_U8) return true;
    if (C == 3 && (eotf_kind == FVVDP_EOTF_SRGB || eotf_kind == FVVDP_EOTF_PQ)) return true;      // uint16 (closed form) / float RGB
    return dtype == FVVDP_F32 && C == 1 && eotf_kind == FVVDP_EOTF_NONE;                          // luminance frames
}
static constexpr int k1_px(int FL, int dtype) {
    return FL == 8 ? (dtype == FVVDP_F32 ? K1_PXF8 : K1_PX8) : (FL == 16 ? K1_PX16 : (FL == 32 ? K1_PX32 : 1));
}

// FL in {8, 16, 32}; dtype FVVDP_U8 / U16 / F32.  FL = 64: see k1_ring64_ok().
void k1_launch_vec(int FL, int dtype, const TemporalArgs& a, hipStream_t st);     // aligned sizes (see the call site)
void k1_launch_ring(int FL, int dtype, const TemporalArgs& a, hipStream_t st);    // any size
void k1_launch_generic(int planes, int dtype, const GenericArgs& a, hipStream_t st);
void k1_launch_luminance(int dtype, const LumArgs& a, hipStream_t st);     // source frames -> fp32 luminance frames (two-pass path)
void k1_launch_yuv_vec(int FL, int bytes, bool c420, bool general_matrix, const YuvArgs& a, hipStream_t st);   // FL in {8, 16}
void k1_launch_yuv(int FL, int bytes, const YuvArgs& a, hipStream_t st);
void k1_launch_yuv_luminance(int bytes, const YuvLumArgs& a, hipStream_t st);   // planar YUV frames -> fp32 luminance frames (two-pass path)
