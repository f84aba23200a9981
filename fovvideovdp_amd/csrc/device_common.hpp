// Device-side helpers shared by the kernels of libfvvdp_hip (included by fvvdp_hip.hip).
#pragma once
// ------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------
template <int P>
struct Pix {
    float v[P];
};

template <int P>
__device__ __forceinline__ Pix<P> ld_pix(const float* p);
template <>
__device__ __forceinline__ Pix<4> ld_pix<4>(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    return Pix<4>{{t.x, t.y, t.z, t.w}};
}
template <>
__device__ __forceinline__ Pix<2> ld_pix<2>(const float* p) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    return Pix<2>{{t.x, t.y}};
}
__device__ __forceinline__ void st_pix(float* p, const Pix<4>& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
__device__ __forceinline__ void st_pix(float* p, const Pix<2>& a) {
    *reinterpret_cast<float2*>(p) = make_float2(a.v[0], a.v[1]);
}

__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Packed fp32 pair: the two video streams (test, reference) or two temporal channels go through the same arithmetic,
// so they are kept as a 2-vector and the compiler emits v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 for them.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
// Stores in straight-line code: a buffer resource covering the destination, and an out-of-range byte offset (dropped by
// the hardware) for lanes or waves that must not write.  See band_kernel.hpp for why this matters (s_waitcnt vmcnt(N)).
#define FVVDP_NO_STORE 0xFFFFFFFFu
__device__ __forceinline__ __amdgpu_buffer_rsrc_t level_rsrc(float* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
// the same for read-only source data (a planar frame of samples)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t src_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// Level 0 of a context lives in ONE range, or in TWO: its even frame slots in one allocation and its odd slots in another, so that the
// temporal kernel's writes fall on both classes of the box's physical memory at any time (place_level0 in fvvdp_hip.hip,
// profiles/r05_k1_mode.md).  One formula serves both: slot s sits at (s odd ? hi : lo) + (s >> 1) * half_stride floats; a single range
// is hi = lo + one frame, half_stride = two frames.  slot0 = absolute slot of the launch's frame 0.
struct L0Addr {
    float* lo;
    float* hi;
    size_t half_stride;
    int slot0;
};
__device__ __forceinline__ float* l0_frame(const L0Addr& a, int f) {
    const int s = a.slot0 + f;
    return ((s & 1) ? a.hi : a.lo) + (size_t)(s >> 1) * a.half_stride;
}
__device__ __forceinline__ v2f splat(float s) { return v2f{s, s}; }
__device__ __forceinline__ v2f pfma(v2f a, float s, v2f c) { return __builtin_elementwise_fma(a, splat(s), c); }
// clamp to [lo, hi] as one v_med3_f32 per component (fminf(fmaxf()) costs three: IEEE max first canonicalises its input)
__device__ __forceinline__ v2f clamp2(v2f x, float lo, float hi) {
    return v2f{__builtin_amdgcn_fmed3f(x.x, lo, hi), __builtin_amdgcn_fmed3f(x.y, lo, hi)};
}
// raw 32-bit value held by the lane to the left / right (0 at the wave's ends): v_mov_b32 with a DPP wave shift
__device__ __forceinline__ unsigned int lane_left_u32(unsigned int x) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned int lane_right_u32(unsigned int x) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}
