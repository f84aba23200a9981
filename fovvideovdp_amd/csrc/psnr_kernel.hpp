// PU21-PSNR side metric: luminance of both streams (same Sampler as the temporal kernels), PU21 encoding and the
// per-frame sum of squared differences.  Reference: pyfvvdp/pupsnr.py:64-79, pyfvvdp/utils.py:183-193.
#pragma once

struct Pu21Args {
    const void* src[2];
    size_t chan_stride, frame_stride;
    int C;
    unsigned int HW;
    unsigned int chunk;       // pixels per slice, multiple of 4
    EotfDev e;
    float w[3];
    float p[7];
    float l_min, l_max;
    double* partial;          // [n_frames][FVVDP_PSNR_SLICES]
    int* oob;
};

// PU.encode (utils.py:183-193).  Powers as exp2(p*log2(x)) on the hardware transcendentals: the first power's error
// is damped by the second exponent (p4 = 0.067), so V carries about twice the rounding noise of an exact fp32
// evaluation (3e-4 at V = 570) -- measured against the reference: < 1e-3 dB.  The quotient is a true division.
__device__ __forceinline__ float pu21_encode(float Y, const Pu21Args& a) {
    Y = __builtin_amdgcn_fmed3f(Y, a.l_min, a.l_max);
    const float Yp = fast_exp2(a.p[3] * fast_log2(Y));
    const float t = (a.p[0] + a.p[1] * Yp) / (1.0f + a.p[2] * Yp);
    return a.p[6] * (fast_exp2(a.p[4] * fast_log2(t)) - a.p[5]);
}

// grid (FVVDP_PSNR_SLICES, n_frames): block (slice, f) sums its pixel range of frame f; fixed thread->pixel mapping
// and fixed reduction order, so the result does not depend on scheduling.
template <int SRC, int PX>
__global__ __launch_bounds__(256) void pu21_sse_kernel(const Pu21Args a_byval) {
    const Pu21Args& a = *(const Pu21Args*)__builtin_amdgcn_kernarg_segment_ptr();      // scalar loads where needed (see band_kernel)
    (void)a_byval;
    __shared__ float lutw[SRC == SRC_U8 ? 768 : 1];
    __shared__ double s_red[4];
    if constexpr (SRC == SRC_U8) {
        build_lutw(lutw, a.e, a.C, a.w, threadIdx.x, 256);
        __syncthreads();
    }
    const float w[3] = {a.C == 3 ? a.w[0] : 1.0f, a.w[1], a.w[2]};
    const unsigned int beg = blockIdx.x * a.chunk;
    const unsigned int end = min(beg + a.chunk, a.HW);
    const size_t foff = (size_t)blockIdx.y * a.frame_stride;
    bool bad = false;
    double acc = 0.0;
    unsigned int p = beg + threadIdx.x * PX;
    // raw samples of the next iteration are requested before the current ones are converted (the conversion is ~130
    // cycles per pixel; without this every iteration started with a full memory round trip)
    RawVecFrame<SRC, PX> n0, n1;
    {
        const size_t o = foff + (p < end ? p : beg);
        n0 = fetch_vec<SRC, PX>(a.src[0], o, a.chan_stride, a.C);
        n1 = fetch_vec<SRC, PX>(a.src[1], o, a.chan_stride, a.C);
    }
    for (; p < end; p += 256 * PX) {
        const RawVecFrame<SRC, PX> c0 = n0, c1 = n1;
        {
            const unsigned int pn = p + 256 * PX;
            const size_t o = foff + (pn < end ? pn : p);          // past the end: the current samples again (unused)
            n0 = fetch_vec<SRC, PX>(a.src[0], o, a.chan_stride, a.C);
            n1 = fetch_vec<SRC, PX>(a.src[1], o, a.chan_stride, a.C);
        }
        float lt[PX], lr[PX];
        frame_lum<SRC, PX, RawVecFrame<SRC, PX>>(c0, a.C, lutw, a.e.lut, w, a.e, lt, bad);
        frame_lum<SRC, PX, RawVecFrame<SRC, PX>>(c1, a.C, lutw, a.e.lut, w, a.e, lr, bad);
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            // The encoded values must be ROUNDED before they are subtracted.  pu21_encode ends in a product; fused into
            // fma(p6, x_t, -(p6 * x_r)) the difference of identical inputs is that product's rounding error instead of 0
            // (the reference returns an infinite PSNR, pupsnr.py:74).  __fmul_rn is a plain `*` in HIP, so the values go
            // through an opaque register move the contraction cannot see through.
            float et = pu21_encode(lt[i], a), er = pu21_encode(lr[i], a);
            asm volatile("" : "+v"(et), "+v"(er));
            const float d = et - er;
            acc += (double)(d * d);              // (img1 - img2)**2 in fp32 like the reference, summed in fp64
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        a.partial[(size_t)blockIdx.y * FVVDP_PSNR_SLICES + blockIdx.x] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
    if (bad && a.oob) atomicOr(a.oob, 1);
}

// one wave per frame: fixed-order sum of the frame's slices
__global__ __launch_bounds__(64) void pu21_finalize_kernel(const double* __restrict__ partial, double* __restrict__ sse) {
    const int f = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < FVVDP_PSNR_SLICES; i += 64) acc += partial[(size_t)f * FVVDP_PSNR_SLICES + i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) sse[f] = acc;
}
