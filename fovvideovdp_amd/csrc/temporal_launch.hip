// Instantiations + launchers of the stage-1 kernels; compiled once per K1_PART (see temporal_launch.hpp).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstddef>
#include <cstdint>
#include <type_traits>

#include "fvvdp_hip.h"
#include "device_common.hpp"
#include "temporal_launch.hpp"

#ifndef K1_PART
#error "compile with -DK1_PART=0..3"
#endif

#if K1_PART <= 2
static constexpr int DT = K1_PART;                     // FVVDP_U8 / FVVDP_U16 / FVVDP_F32
static constexpr int SRC = DT == FVVDP_U8 ? SRC_U8 : (DT == FVVDP_U16 ? SRC_U16 : SRC_F32);

template <int FL>
static void launch_vec(const TemporalArgs& a, hipStream_t st) {
    constexpr int PX = k1_px(FL, DT);
    constexpr int TD = FL == 8 ? K1_TD8 : (FL == 16 ? K1_TD16 : (FL == 32 ? K1_TD32 : 1));
    constexpr int WPB = k1_wpb(FL);                       // waves per workgroup, one pixel block each
    const int n_blocks = (a.HW + 64 * PX - 1) / (64 * PX);
    dim3 grid((n_blocks + WPB - 1) / WPB), block(64 * WPB);
    if (a.ticket) {          // resident workgroups that take their pixel blocks from a.ticket (uint8, FL <= 16; the caller zeroed it)
        static int resident = 0;
        if (!resident) {
            int dev = 0, per_cu = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, temporal_vec_kernel<FL, PX, SRC, TD>, 64, 0) == hipSuccess && per_cu > 0)
                resident = per_cu * prop.multiProcessorCount;
            else { (void)hipGetLastError(); resident = -1; }
        }
        if (resident > 0 && (int)grid.x > 2 * resident) grid.x = (unsigned int)resident;
        else { TemporalArgs b = a; b.ticket = nullptr; hipLaunchKernelGGL((temporal_vec_kernel<FL, PX, SRC, TD>), grid, block, 0, st, b); return; }
    }
    hipLaunchKernelGGL((temporal_vec_kernel<FL, PX, SRC, TD>), grid, block, 0, st, a);
}
template <int FL, int PX>
static void launch_ring(const TemporalArgs& a, hipStream_t st) {
    dim3 grid((a.HW + 256 * PX - 1) / (256 * PX)), block(256);
    hipLaunchKernelGGL((temporal_ring_kernel<FL, PX, SRC>), grid, block, 0, st, a);
}
template <int P>
static void launch_generic(const GenericArgs& a, hipStream_t st) {
    dim3 grid((a.HW + 255) / 256, a.n_out), block(256);
    hipLaunchKernelGGL((temporal_generic_kernel<SRC, P>), grid, block, 0, st, a);
}

#if K1_PART == 0
#define K1_NAME(f) f##_u8
#elif K1_PART == 1
#define K1_NAME(f) f##_u16
#else
#define K1_NAME(f) f##_f32
#endif
#if defined(K1_TIMELINE) && K1_PART == 0      // profiling build only: not part of the C ABI
extern "C" int fvvdp_debug_k1_timeline(unsigned long long* h_out, size_t n_records) {
    if (n_records > 65536) n_records = 65536;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_k1_timeline), n_records * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
void K1_NAME(k1_vec)(int FL, const TemporalArgs& a, hipStream_t st) {
    if (FL == 8) launch_vec<8>(a, st);
    else if (FL == 16) launch_vec<16>(a, st);
    else if (FL == 64) launch_vec<64>(a, st);     // the cases of k1_ring64_ok() only (compile time: the loop is unrolled FL times per case)
    else launch_vec<32>(a, st);
}
void K1_NAME(k1_ring)(int FL, const TemporalArgs& a, hipStream_t st) {
    if (FL == 8) launch_ring<8, 4>(a, st);
    else if (FL == 16) launch_ring<16, 4>(a, st);
    else if (FL == 32) launch_ring<32, 2>(a, st);
    // FL == 64 has no per-pixel ring instantiation: the caller (temporal_channels_core) refuses it before it gets here
}
void K1_NAME(k1_generic)(int planes, const GenericArgs& a, hipStream_t st) {
    if (planes == 2) launch_generic<2>(a, st);
    else launch_generic<4>(a, st);
}
void K1_NAME(k1_luminance)(const LumArgs& a, hipStream_t st) {
    dim3 grid((a.HW + 255) / 256, a.n_frames), block(256);
    hipLaunchKernelGGL((luminance_frames_kernel<SRC>), grid, block, 0, st, a);
}
#endif

#if K1_PART == 3
void k1_vec_u8(int, const TemporalArgs&, hipStream_t);
void k1_vec_u16(int, const TemporalArgs&, hipStream_t);
void k1_vec_f32(int, const TemporalArgs&, hipStream_t);
void k1_ring_u8(int, const TemporalArgs&, hipStream_t);
void k1_ring_u16(int, const TemporalArgs&, hipStream_t);
void k1_ring_f32(int, const TemporalArgs&, hipStream_t);
void k1_generic_u8(int, const GenericArgs&, hipStream_t);
void k1_generic_u16(int, const GenericArgs&, hipStream_t);
void k1_generic_f32(int, const GenericArgs&, hipStream_t);
void k1_luminance_u8(const LumArgs&, hipStream_t);
void k1_luminance_u16(const LumArgs&, hipStream_t);
void k1_luminance_f32(const LumArgs&, hipStream_t);

void k1_launch_vec(int FL, int dtype, const TemporalArgs& a, hipStream_t st) {
    if (dtype == FVVDP_U8) k1_vec_u8(FL, a, st);
    else if (dtype == FVVDP_U16) k1_vec_u16(FL, a, st);
    else k1_vec_f32(FL, a, st);
}
void k1_launch_ring(int FL, int dtype, const TemporalArgs& a, hipStream_t st) {
    if (dtype == FVVDP_U8) k1_ring_u8(FL, a, st);
    else if (dtype == FVVDP_U16) k1_ring_u16(FL, a, st);
    else k1_ring_f32(FL, a, st);
}
void k1_launch_luminance(int dtype, const LumArgs& a, hipStream_t st) {
    if (dtype == FVVDP_U8) k1_luminance_u8(a, st);
    else if (dtype == FVVDP_U16) k1_luminance_u16(a, st);
    else k1_luminance_f32(a, st);
}
void k1_launch_generic(int planes, int dtype, const GenericArgs& a, hipStream_t st) {
    if (dtype == FVVDP_U8) k1_generic_u8(planes, a, st);
    else if (dtype == FVVDP_U16) k1_generic_u16(planes, a, st);
    else k1_generic_f32(planes, a, st);
}

template <int FL, int PX>
static void launch_yuv(int bytes, const YuvArgs& a, hipStream_t st) {
    const int HW = a.W * a.H;
    dim3 grid((HW + 256 * PX - 1) / (256 * PX)), block(256);
    if (bytes == 1) hipLaunchKernelGGL((temporal_yuv_kernel<FL, PX, unsigned char>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((temporal_yuv_kernel<FL, PX, unsigned short>), grid, block, 0, st, a);
}
// the colour matrix has the shape of the ITU YCbCr matrices (temporal_kernels.hpp, yuv_pair_rgb<.., STDM>)
static bool yuv_matrix_is_standard(const YuvArgs& a) {
    return a.m[0] == 1.0f && a.m[3] == 1.0f && a.m[6] == 1.0f && a.m[1] == 0.0f && a.m[8] == 0.0f;
}
template <int FL, typename T, bool C420, bool STDM>
static void launch_yuv_vec_kind(const YuvArgs& a, dim3 grid, hipStream_t st) {
    constexpr int WPB = yuv_wpb(FL);                       // waves per workgroup, one run of 62 pixel quads each
    const dim3 block(64 * WPB);
    grid.x = (grid.x + WPB - 1) / WPB;
    switch (a.e.kind) {          // the display model is a template constant of the kernel
        case FVVDP_EOTF_SRGB: hipLaunchKernelGGL((temporal_yuv_vec_kernel<FL, T, C420, FVVDP_EOTF_SRGB, STDM>), grid, block, 0, st, a); break;
        case FVVDP_EOTF_GAMMA: hipLaunchKernelGGL((temporal_yuv_vec_kernel<FL, T, C420, FVVDP_EOTF_GAMMA, STDM>), grid, block, 0, st, a); break;
        case FVVDP_EOTF_PQ: hipLaunchKernelGGL((temporal_yuv_vec_kernel<FL, T, C420, FVVDP_EOTF_PQ, STDM>), grid, block, 0, st, a); break;
        case FVVDP_EOTF_LINEAR: hipLaunchKernelGGL((temporal_yuv_vec_kernel<FL, T, C420, FVVDP_EOTF_LINEAR, STDM>), grid, block, 0, st, a); break;
        case FVVDP_EOTF_ABSOLUTE: hipLaunchKernelGGL((temporal_yuv_vec_kernel<FL, T, C420, FVVDP_EOTF_ABSOLUTE, STDM>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((temporal_yuv_vec_kernel<FL, T, C420, FVVDP_EOTF_NONE, STDM>), grid, block, 0, st, a); break;
    }
}
template <int FL, typename T, bool C420>
static void launch_yuv_vec_kind(bool general, const YuvArgs& a, dim3 grid, hipStream_t st) {
    if (yuv_matrix_is_standard(a) && !general) launch_yuv_vec_kind<FL, T, C420, true>(a, grid, st);
    else launch_yuv_vec_kind<FL, T, C420, false>(a, grid, st);
}
template <int FL>
static void launch_yuv_vec(int bytes, bool c420, bool general, const YuvArgs& a, hipStream_t st) {
    const int HW = a.W * a.H;
    dim3 grid((HW / yuv_px(FL) + YUV_QUADS - 1) / YUV_QUADS);      // one wave per run of 62 pixel groups
    if (bytes == 1) {
        if (c420) launch_yuv_vec_kind<FL, unsigned char, true>(general, a, grid, st);
        else launch_yuv_vec_kind<FL, unsigned char, false>(general, a, grid, st);
    } else {
        if (c420) launch_yuv_vec_kind<FL, unsigned short, true>(general, a, grid, st);
        else launch_yuv_vec_kind<FL, unsigned short, false>(general, a, grid, st);
    }
}
void k1_launch_yuv_vec(int FL, int bytes, bool c420, bool general_matrix, const YuvArgs& a, hipStream_t st) {
    if (FL == 8) launch_yuv_vec<8>(bytes, c420, general_matrix, a, st);
    else launch_yuv_vec<16>(bytes, c420, general_matrix, a, st);
}
void k1_launch_yuv_luminance(int bytes, const YuvLumArgs& a, hipStream_t st) {
    dim3 grid((a.y.W * a.y.H + 255) / 256, a.n_frames), block(256);
    if (bytes == 1) hipLaunchKernelGGL((yuv_luminance_frames_kernel<unsigned char>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((yuv_luminance_frames_kernel<unsigned short>), grid, block, 0, st, a);
}
void k1_launch_yuv(int FL, int bytes, const YuvArgs& a, hipStream_t st) {
    if (FL == 8) launch_yuv<8, 2>(bytes, a, st);
    else if (FL == 16) launch_yuv<16, 2>(bytes, a, st);
    else launch_yuv<32, 1>(bytes, a, st);
}
#endif
