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

