// libfvvdp_hip: FovVideoVDP per-frame visible-difference path for MI355X (gfx950 / CDNA4).
//
// Written for gfx950 only: wave64, single-wave workgroups that stream down the image for the fused pyramid
// kernel, 16-byte-per-lane coalesced HBM access on pixel-interleaved planes, LDS for neighbour exchange, no MFMA
// (5-tap stencils + pointwise + LUT: there is no contraction to feed a matrix core).  See DESIGN.md.
//
// Data layout in HBM (context scratch): Gaussian level i of frame slot s is an array [h_i][w_i][P] fp32 --
// the P temporal-channel planes of one pixel are adjacent (one aligned float4 for video, float2 for images), so
// one lane = one pixel, every load/store is 16 B (8 B) wide and all per-pixel math is thread-local.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fvvdp_hip.h"

// ------------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(FVVDP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* fvvdp_last_error(void) { return g_err; }

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

// ------------------------------------------------------------------------------------------------------------
// stage 1: unpack + display photometry + luminance + temporal FIR  ->  pyramid level 0 (interleaved planes)
// ------------------------------------------------------------------------------------------------------------
enum { SRC_U8 = 0, SRC_U16 = 1, SRC_F32 = 2 };

struct EotfDev {
    int kind;
    float scale;    // Y_peak - Y_black
    float y_black;
    float y_peak;
    float gamma;
    float l_min, l_max;
    const float* lut;
};

// Per-channel display model on a float sample V (fvvdp_display_model.py:147-165).  `bad` is set when V was
// outside [0,1] for an EOTF that clamps.
__device__ __forceinline__ float eotf_f32(float V, const EotfDev& e, bool& bad) {
    switch (e.kind) {
        case FVVDP_EOTF_SRGB: {
            bad = bad || (V > 1.0f) || (V < 0.0f);
            V = fminf(fmaxf(V, 0.0f), 1.0f);
            const float hi = fast_exp2(2.4f * fast_log2((V + 0.055f) / 1.055f));
            const float lin = V > 0.04045f ? hi : V / 12.92f;
            return __fadd_rn(__fmul_rn(e.scale, lin), e.y_black);
        }
        case FVVDP_EOTF_GAMMA: {
            bad = bad || (V > 1.0f) || (V < 0.0f);
            V = fminf(fmaxf(V, 0.0f), 1.0f);
            const float lin = V > 0.0f ? fast_exp2(e.gamma * fast_log2(V)) : 0.0f;
            return __fadd_rn(__fmul_rn(e.scale, lin), e.y_black);
        }
        case FVVDP_EOTF_PQ: {
            bad = bad || (V > 1.0f) || (V < 0.0f);
            V = fminf(fmaxf(V, 0.0f), 1.0f);
            const float m = 78.843750000000000f, n = 0.15930175781250000f;
            const float c1 = 0.83593750000000000f, c2 = 18.851562500000000f, c3 = 18.687500000000000f;
            const float im_t = V > 0.0f ? fast_exp2(fast_log2(V) * (1.0f / m)) : 0.0f;
            const float r = fmaxf(im_t - c1, 0.0f) / (c2 - c3 * im_t);
            const float L = r > 0.0f ? 10000.0f * fast_exp2(fast_log2(r) * (1.0f / n)) : 0.0f;
            return fminf(fmaxf(L, 0.005f), e.y_peak) + e.y_black;
        }
        case FVVDP_EOTF_LINEAR:
            return fminf(fmaxf(V, 0.005f), e.y_peak) + e.y_black;
        case FVVDP_EOTF_ABSOLUTE:
            return fminf(fmaxf(V, e.l_min), e.l_max);
        default:
            return V;
    }
}

// Luminance of PX consecutive pixels of one frame of one stream.
//   U8 : LDS table lutw[c][code] = lut[code]*w[c] (same products and same summation order as the reference:
//        (Lr*w0 + Lg*w1) + Lb*w2, video_source.py:206)
template <int SRC, int PX>
struct Sampler {
    const void* base;
    size_t chan_stride;
    int C;
    const float* lutw;     // LDS, [3][256], SRC_U8 only
    const float* lut16;    // global, SRC_U16 only
    float w0, w1, w2;
    EotfDev e;

    __device__ __forceinline__ void chan(const void* p, size_t off, float (&o)[PX], int c, bool& bad) const {
        if constexpr (SRC == SRC_U8) {
            const unsigned char* q = reinterpret_cast<const unsigned char*>(p) + off;
            unsigned char code[PX];
            if constexpr (PX == 4) {
                const uchar4 t = *reinterpret_cast<const uchar4*>(q);
                code[0] = t.x; code[1] = t.y; code[2] = t.z; code[3] = t.w;
            } else if constexpr (PX == 2) {
                const uchar2 t = *reinterpret_cast<const uchar2*>(q);
                code[0] = t.x; code[1] = t.y;
            } else {
                code[0] = *q;
            }
#pragma unroll
            for (int i = 0; i < PX; ++i) o[i] = lutw[c * 256 + code[i]];
        } else if constexpr (SRC == SRC_U16) {
            const unsigned short* q = reinterpret_cast<const unsigned short*>(p) + off;
            unsigned short code[PX];
            if constexpr (PX == 4) {
                const ushort4 t = *reinterpret_cast<const ushort4*>(q);
                code[0] = t.x; code[1] = t.y; code[2] = t.z; code[3] = t.w;
            } else if constexpr (PX == 2) {
                const ushort2 t = *reinterpret_cast<const ushort2*>(q);
                code[0] = t.x; code[1] = t.y;
            } else {
                code[0] = *q;
            }
            const float wc = (c == 0) ? w0 : ((c == 1) ? w1 : w2);
#pragma unroll
            for (int i = 0; i < PX; ++i) o[i] = __fmul_rn(lut16[code[i]], wc);
        } else {
            const float* q = reinterpret_cast<const float*>(p) + off;
            float V[PX];
            if constexpr (PX == 4) {
                const float4 t = *reinterpret_cast<const float4*>(q);
                V[0] = t.x; V[1] = t.y; V[2] = t.z; V[3] = t.w;
            } else if constexpr (PX == 2) {
                const float2 t = *reinterpret_cast<const float2*>(q);
                V[0] = t.x; V[1] = t.y;
            } else {
                V[0] = *q;
            }
            const float wc = (c == 0) ? w0 : ((c == 1) ? w1 : w2);
#pragma unroll
            for (int i = 0; i < PX; ++i) o[i] = __fmul_rn(eotf_f32(V[i], e, bad), wc);
        }
    }

    // frame offset `foff` (elements) already includes f*frame_stride + pixel index
    __device__ __forceinline__ void lum(size_t foff, float (&L)[PX], bool& bad) const {
        if (C == 3) {
            float a[PX], b[PX], c[PX];
            chan(base, foff, a, 0, bad);
            chan(base, foff + chan_stride, b, 1, bad);
            chan(base, foff + 2 * chan_stride, c, 2, bad);
#pragma unroll
            for (int i = 0; i < PX; ++i) L[i] = __fadd_rn(__fadd_rn(a[i], b[i]), c[i]);
        } else {
            chan(base, foff, L, 0, bad);
        }
    }
};

#define T_MAX_IDX 320   // history + outputs of one launch
struct TemporalArgs {
    const void* src[2];
    size_t chan_stride, frame_stride;
    int C, HW;
    EotfDev e;
    float w[3];
    int n_out;
    int fl;                // true filter length (<= FL)
    float* out;            // level 0 of the first output slot: [n_out][HW][4]
    int* oob;
    float taps[2][32];
    int idx[T_MAX_IDX];    // [FL-1+n_out], entries before the true history are padded with a valid frame
};

__device__ __forceinline__ void build_lutw(float* lutw, const float* lut, int C, const float* w, int tid, int nthreads) {
    for (int i = tid; i < 256; i += nthreads) {
        const float l = lut[i];
        if (C == 3) {
            lutw[i] = __fmul_rn(l, w[0]);
            lutw[256 + i] = __fmul_rn(l, w[1]);
            lutw[512 + i] = __fmul_rn(l, w[2]);
        } else {
            lutw[i] = l;
        }
    }
}

// Temporally tiled FIR: one thread owns PX pixels for the whole launch and keeps the last FL luminance values of
// both streams in registers (ring with compile-time slot indices), so every source frame is read exactly once and
// every output pixel is written once as one float4 (test-sust, ref-sust, test-trans, ref-trans).
// Reference: fvvdp.py:294-300 (R[:,2cc+s] = sum_k window[s][k] * F[cc].flip(0)[k]).
// PX raw samples of one channel.  A thread's PX pixels are 256 apart (pixel i of thread t in block b is
// b*256*PX + i*256 + t): every load and every store instruction of a wave then covers one contiguous run of
// memory (64 x 1/2/4 B loads, 64 x 16 B = 1 KiB stores of finished float4 pixels).
template <int SRC, int PX>
struct RawPx {
    unsigned int wd[PX];
    __device__ __forceinline__ unsigned int code(int i) const { return wd[i]; }              // integer sources
    __device__ __forceinline__ float value(int i) const { return __uint_as_float(wd[i]); }   // float source
};

template <int SRC, int PX>
__device__ __forceinline__ RawPx<SRC, PX> load_raw(const void* base, size_t off, const int (&px)[PX]) {
    RawPx<SRC, PX> r;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        if constexpr (SRC == SRC_U8) r.wd[i] = reinterpret_cast<const unsigned char*>(base)[off + px[i]];
        else if constexpr (SRC == SRC_U16) r.wd[i] = reinterpret_cast<const unsigned short*>(base)[off + px[i]];
        else r.wd[i] = reinterpret_cast<const unsigned int*>(base)[off + px[i]];
    }
    return r;
}

// raw samples of all channels of one frame of one stream -> luminance of PX pixels
template <int SRC, int PX>
struct RawFrame {
    RawPx<SRC, PX> ch[3];
};

template <int SRC, int PX>
__device__ __forceinline__ RawFrame<SRC, PX> fetch_frame(const void* base, size_t off, size_t chan_stride, int C,
                                                         const int (&px)[PX]) {
    RawFrame<SRC, PX> f;
    f.ch[0] = load_raw<SRC, PX>(base, off, px);
    if (C == 3) {
        f.ch[1] = load_raw<SRC, PX>(base, off + chan_stride, px);
        f.ch[2] = load_raw<SRC, PX>(base, off + 2 * chan_stride, px);
    } else {
        f.ch[1] = f.ch[0];
        f.ch[2] = f.ch[0];
    }
    return f;
}

template <int SRC, int PX, typename FRAME>
__device__ __forceinline__ void frame_lum(const FRAME& f, int C, const float* lutw, const float* lut16,
                                          const float (&w)[3], const EotfDev& e, float (&L)[PX], bool& bad) {
    float v[3][PX];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c > 0 && C != 3) break;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            if constexpr (SRC == SRC_U8) v[c][i] = lutw[c * 256 + f.ch[c].code(i)];
            else if constexpr (SRC == SRC_U16) v[c][i] = __fmul_rn(lut16[f.ch[c].code(i)], w[c]);
            else v[c][i] = __fmul_rn(eotf_f32(f.ch[c].value(i), e, bad), w[c]);
        }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) L[i] = (C == 3) ? __fadd_rn(__fadd_rn(v[0][i], v[1][i]), v[2][i]) : v[0][i];
}

// Temporally tiled FIR: one thread owns PX pixels for the whole launch and keeps the last FL luminance values of
// both streams in registers (ring with compile-time slot indices), so every source frame is read exactly once and
// every output pixel is written once as one float4 (test-sust, ref-sust, test-trans, ref-trans).  The raw samples
// of the next frame are fetched while the current one is filtered (software prefetch, one frame ahead).
// Reference: fvvdp.py:294-300 (R[:,2cc+s] = sum_k window[s][k] * F[cc].flip(0)[k]).
#ifndef TDIST
#define TDIST 1          // frames of raw samples in flight per thread; 2 and 4 measured slower (VGPRs -> occupancy)
#endif
template <int FL, int PX, int SRC>
__global__ __launch_bounds__(256) void temporal_ring_kernel(const TemporalArgs a) {
    __shared__ float lutw[SRC == SRC_U8 ? 768 : 1];
    if constexpr (SRC == SRC_U8) {
        build_lutw(lutw, a.e.lut, a.C, a.w, threadIdx.x, 256);
        __syncthreads();
    }
    int px[PX];          // this thread's pixels (clamped for the loads; stores are predicated on `ok`)
    bool ok[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int q = blockIdx.x * (256 * PX) + i * 256 + threadIdx.x;
        ok[i] = q < a.HW;
        px[i] = ok[i] ? q : a.HW - 1;
    }
    const float w[3] = {a.C == 3 ? a.w[0] : 1.0f, a.w[1], a.w[2]};
    bool bad = false;
    float ring[2][FL][PX];
#pragma unroll
    for (int u = 0; u < FL; ++u)
#pragma unroll
        for (int i = 0; i < PX; ++i) ring[0][u][i] = ring[1][u][i] = 0.0f;
    // virtual time v = 0 .. FL-2 is the history, v = FL-1+t the newest frame of output t; ring slot = v % FL.
    // The same pipelined loop fills the history and produces the outputs, so at most one frame is in flight.
    const int total = FL - 1 + a.n_out;
    RawFrame<SRC, PX> nx[TDIST][2];           // raw samples of the next TDIST frames, in flight
#pragma unroll
    for (int d = 0; d < TDIST; ++d) {
        const size_t off = (size_t)a.idx[d < total ? d : total - 1] * a.frame_stride;
        nx[d][0] = fetch_frame<SRC, PX>(a.src[0], off, a.chan_stride, a.C, px);
        nx[d][1] = fetch_frame<SRC, PX>(a.src[1], off, a.chan_stride, a.C, px);
    }
    for (int v0 = 0; v0 < total; v0 += FL) {
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int v = v0 + u;
            if (v < total) {
                const RawFrame<SRC, PX> cur0 = nx[u % TDIST][0], cur1 = nx[u % TDIST][1];
                if (v + TDIST < total) {
                    const size_t off = (size_t)a.idx[v + TDIST] * a.frame_stride;
                    nx[u % TDIST][0] = fetch_frame<SRC, PX>(a.src[0], off, a.chan_stride, a.C, px);
                    nx[u % TDIST][1] = fetch_frame<SRC, PX>(a.src[1], off, a.chan_stride, a.C, px);
                }
                frame_lum<SRC, PX, RawFrame<SRC, PX>>(cur0, a.C, lutw, a.e.lut, w, a.e, ring[0][u], bad);
                frame_lum<SRC, PX, RawFrame<SRC, PX>>(cur1, a.C, lutw, a.e.lut, w, a.e, ring[1][u], bad);
                if (v >= FL - 1) {
                    float acc[4][PX];
#pragma unroll
                    for (int i = 0; i < PX; ++i) acc[0][i] = acc[1][i] = acc[2][i] = acc[3][i] = 0.0f;
                    // oldest tap first, like the reference's sum over the window dimension
#pragma unroll
                    for (int k = FL - 1; k >= 0; --k) {
                        const int sl = (u - k + 2 * FL) % FL;
                        const float f0 = a.taps[0][k], f1 = a.taps[1][k];
#pragma unroll
                        for (int i = 0; i < PX; ++i) {
                            acc[0][i] = fmaf(ring[0][sl][i], f0, acc[0][i]);
                            acc[1][i] = fmaf(ring[1][sl][i], f0, acc[1][i]);
                            acc[2][i] = fmaf(ring[0][sl][i], f1, acc[2][i]);
                            acc[3][i] = fmaf(ring[1][sl][i], f1, acc[3][i]);
                        }
                    }
                    float* o = a.out + (size_t)(v - (FL - 1)) * a.HW * 4;
#pragma unroll
                    for (int i = 0; i < PX; ++i)
                        if (ok[i])
                            *reinterpret_cast<float4*>(o + (size_t)px[i] * 4) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
                }
            }
        }
    }
    if (bad && a.oob) atomicOr(a.oob, 1);
}

// ---- vector variant of the temporally tiled FIR (the fast path) -------------------------------------------------
// Single-wave workgroups; a lane owns PX CONSECUTIVE pixels, so one load per channel fetches all of them
// (4 uint8 = one dword, 4 uint16 = 8 B, 4 fp32 = 16 B: 4x fewer memory instructions than the per-pixel loads of
// temporal_ring_kernel).  The finished float4 pixels are transposed through LDS (padded rows, conflict-free) so that
// every store instruction of the wave still writes one contiguous 1 KiB run.  Needs HW % PX == 0 and PX-sample
// aligned strides; other sizes take temporal_ring_kernel.
template <int SRC, int PX>
struct RawVec {
    static constexpr int ES = (SRC == SRC_U8 ? 1 : (SRC == SRC_U16 ? 2 : 4));
    static constexpr int WORDS = (ES * PX + 3) / 4;
    unsigned int wd[WORDS];
    __device__ __forceinline__ unsigned int code(int i) const {
        if constexpr (SRC == SRC_U8) return (wd[i / 4] >> (8 * (i % 4))) & 0xFFu;
        else return (wd[i / 2] >> (16 * (i % 2))) & 0xFFFFu;
    }
    __device__ __forceinline__ float value(int i) const { return __uint_as_float(wd[i]); }
};
template <int SRC, int PX>
struct RawVecFrame {
    RawVec<SRC, PX> ch[3];
};

template <int SRC, int PX>
__device__ __forceinline__ RawVec<SRC, PX> load_vec(const void* base, size_t off) {
    RawVec<SRC, PX> r;
    constexpr int ES = RawVec<SRC, PX>::ES;
    constexpr int BYTES = ES * PX;
    const char* q = reinterpret_cast<const char*>(base) + off * ES;
    if constexpr (BYTES == 16) {
        const uint4 t = *reinterpret_cast<const uint4*>(q);
        r.wd[0] = t.x; r.wd[1] = t.y; r.wd[2] = t.z; r.wd[3] = t.w;
    } else if constexpr (BYTES == 8) {
        const uint2 t = *reinterpret_cast<const uint2*>(q);
        r.wd[0] = t.x; r.wd[1] = t.y;
    } else if constexpr (BYTES == 4) {
        r.wd[0] = *reinterpret_cast<const unsigned int*>(q);
    } else {
        r.wd[0] = *reinterpret_cast<const unsigned short*>(q);
    }
    return r;
}

template <int SRC, int PX>
__device__ __forceinline__ RawVecFrame<SRC, PX> fetch_vec(const void* base, size_t off, size_t chan_stride, int C) {
    RawVecFrame<SRC, PX> f;
    f.ch[0] = load_vec<SRC, PX>(base, off);
    if (C == 3) {
        f.ch[1] = load_vec<SRC, PX>(base, off + chan_stride);
        f.ch[2] = load_vec<SRC, PX>(base, off + 2 * chan_stride);
    } else {
        f.ch[1] = f.ch[0];
        f.ch[2] = f.ch[0];
    }
    return f;
}

template <int FL, int PX, int SRC>
__global__ __launch_bounds__(64) void temporal_vec_kernel(const TemporalArgs a) {
    __shared__ float lutw[SRC == SRC_U8 ? 768 : 1];
    __shared__ float4 s_t[64 * (PX + 1)];          // one padded row of PX float4 per lane
    const int lane = threadIdx.x;
    if constexpr (SRC == SRC_U8) build_lutw(lutw, a.e.lut, a.C, a.w, lane, 64);
    __syncthreads();
    const int p0 = blockIdx.x * (64 * PX);          // first pixel of this wave
    const int pl = min(p0 + lane * PX, a.HW - PX);  // this lane's PX consecutive pixels (clamped: loads stay in range)
    const float w[3] = {a.C == 3 ? a.w[0] : 1.0f, a.w[1], a.w[2]};
    bool bad = false;
    float ring[2][FL][PX];
#pragma unroll
    for (int u = 0; u < FL; ++u)
#pragma unroll
        for (int i = 0; i < PX; ++i) ring[0][u][i] = ring[1][u][i] = 0.0f;
    const int total = FL - 1 + a.n_out;
    RawVecFrame<SRC, PX> nx[2];
    {
        const size_t off = (size_t)a.idx[0] * a.frame_stride + pl;
        nx[0] = fetch_vec<SRC, PX>(a.src[0], off, a.chan_stride, a.C);
        nx[1] = fetch_vec<SRC, PX>(a.src[1], off, a.chan_stride, a.C);
    }
    for (int v0 = 0; v0 < total; v0 += FL) {
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int v = v0 + u;
            if (v < total) {
                const RawVecFrame<SRC, PX> cur0 = nx[0], cur1 = nx[1];
                if (v + 1 < total) {
                    const size_t off = (size_t)a.idx[v + 1] * a.frame_stride + pl;
                    nx[0] = fetch_vec<SRC, PX>(a.src[0], off, a.chan_stride, a.C);
                    nx[1] = fetch_vec<SRC, PX>(a.src[1], off, a.chan_stride, a.C);
                }
                frame_lum<SRC, PX, RawVecFrame<SRC, PX>>(cur0, a.C, lutw, a.e.lut, w, a.e, ring[0][u], bad);
                frame_lum<SRC, PX, RawVecFrame<SRC, PX>>(cur1, a.C, lutw, a.e.lut, w, a.e, ring[1][u], bad);
                if (v >= FL - 1) {
                    float acc[4][PX];
#pragma unroll
                    for (int i = 0; i < PX; ++i) acc[0][i] = acc[1][i] = acc[2][i] = acc[3][i] = 0.0f;
#pragma unroll
                    for (int k = FL - 1; k >= 0; --k) {      // oldest tap first, like the reference's sum over the window
                        const int sl = (u - k + 2 * FL) % FL;
                        const float f0 = a.taps[0][k], f1 = a.taps[1][k];
#pragma unroll
                        for (int i = 0; i < PX; ++i) {
                            acc[0][i] = fmaf(ring[0][sl][i], f0, acc[0][i]);
                            acc[1][i] = fmaf(ring[1][sl][i], f0, acc[1][i]);
                            acc[2][i] = fmaf(ring[0][sl][i], f1, acc[2][i]);
                            acc[3][i] = fmaf(ring[1][sl][i], f1, acc[3][i]);
                        }
                    }
                    // transpose through LDS: lane l holds pixels l*PX..l*PX+PX-1, store i writes pixels i*64+l
                    __syncthreads();                         // single wave: orders the LDS accesses only
#pragma unroll
                    for (int i = 0; i < PX; ++i)
                        s_t[lane * (PX + 1) + i] = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
                    __syncthreads();
                    float4* o = reinterpret_cast<float4*>(a.out) + (size_t)(v - (FL - 1)) * a.HW + p0;
#pragma unroll
                    for (int i = 0; i < PX; ++i) {
                        const int q = i * 64 + lane;
                        const float4 val = s_t[(q / PX) * (PX + 1) + (q % PX)];
                        if (p0 + q < a.HW) o[q] = val;
                    }
                }
            }
        }
    }
    if (bad && a.oob) atomicOr(a.oob, 1);
}

// ---- planar YUV ingest fused with the temporal filter ---------------------------------------------------------
// Replaces video_reader_yuv_pytorch.unpack / _fixed2float_upscale (video_source_file.py:219-276) + _prepare_frame
// (:355-363): limited-range fixed->float (Y: w*Y-16/219 clipped to [0,1]; Cb,Cr: w*c-128/224 clipped to +-0.5),
// 4:2:0 chroma bilinear x2 (torch interpolate, align_corners=False: source = (dst+0.5)/2-0.5 clamped at 0),
// YCbCr->RGB matrix, clip to [0,1], display model per channel, RGB->luminance, then the same register-ring FIR
// as the other temporal kernels.  One thread owns PX pixels 256 apart (coalesced Y loads and float4 stores).
struct YuvArgs {
    const void* src[2];
    size_t frame_stride;     // elements between frames
    int W, H, uvw, uvh;
    int chroma420;
    float wy, wc;            // 1/(2^(b-8)*219), 1/(2^(b-8)*224)
    float m[9];              // ycbcr2rgb, row-major: R = m0*Y + m1*Cb + m2*Cr ...
    EotfDev e;
    float w[3];
    int n_out, fl;
    float* out;
    int* oob;
    float taps[2][32];
    int idx[T_MAX_IDX];
};

template <typename T>
__device__ __forceinline__ float yuv_lum(const T* __restrict__ f, const YuvArgs& a, int p, bool& bad) {
    const int HW = a.W * a.H;
    const int y = p / a.W, x = p - y * a.W;
    const float Yf = fminf(fmaxf(a.wy * (float)f[p] - (16.0f / 219.0f), 0.0f), 1.0f);
    const T* U = f + HW;
    const T* V = U + a.uvw * a.uvh;
    auto cf = [&](const T* pl, int yy, int xx) {
        return fminf(fmaxf(a.wc * (float)pl[yy * a.uvw + xx] - (128.0f / 224.0f), -0.5f), 0.5f);
    };
    float u, v;
    if (a.chroma420) {
        const float sy = fmaxf(((float)y + 0.5f) * 0.5f - 0.5f, 0.0f), sx = fmaxf(((float)x + 0.5f) * 0.5f - 0.5f, 0.0f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, a.uvh - 1), x1 = min(x0 + 1, a.uvw - 1);
        const float fy = sy - (float)y0, fx = sx - (float)x0;
        const float gy = 1.0f - fy, gx = 1.0f - fx;
        u = gy * (gx * cf(U, y0, x0) + fx * cf(U, y0, x1)) + fy * (gx * cf(U, y1, x0) + fx * cf(U, y1, x1));
        v = gy * (gx * cf(V, y0, x0) + fx * cf(V, y0, x1)) + fy * (gx * cf(V, y1, x0) + fx * cf(V, y1, x1));
    } else {
        u = cf(U, y, x);
        v = cf(V, y, x);
    }
    float L = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float rgb = a.m[3 * c] * Yf + a.m[3 * c + 1] * u + a.m[3 * c + 2] * v;
        rgb = fminf(fmaxf(rgb, 0.0f), 1.0f);
        const float l = __fmul_rn(eotf_f32(rgb, a.e, bad), a.w[c]);
        L = (c == 0) ? l : __fadd_rn(L, l);
    }
    return L;
}

template <int FL, int PX, typename T>
__global__ __launch_bounds__(256) void temporal_yuv_kernel(const YuvArgs a) {
    const int HW = a.W * a.H;
    int px[PX];
    bool ok[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int q = blockIdx.x * (256 * PX) + i * 256 + threadIdx.x;
        ok[i] = q < HW;
        px[i] = ok[i] ? q : HW - 1;
    }
    bool bad = false;
    float ring[2][FL][PX];
#pragma unroll
    for (int u = 0; u < FL; ++u)
#pragma unroll
        for (int i = 0; i < PX; ++i) ring[0][u][i] = ring[1][u][i] = 0.0f;
    const int total = FL - 1 + a.n_out;
    for (int v0 = 0; v0 < total; v0 += FL) {
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int v = v0 + u;
            if (v < total) {
                const size_t off = (size_t)a.idx[v] * a.frame_stride;
                const T* f0 = reinterpret_cast<const T*>(a.src[0]) + off;
                const T* f1 = reinterpret_cast<const T*>(a.src[1]) + off;
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    ring[0][u][i] = yuv_lum<T>(f0, a, px[i], bad);
                    ring[1][u][i] = yuv_lum<T>(f1, a, px[i], bad);
                }
                if (v >= FL - 1) {
                    float acc[4][PX];
#pragma unroll
                    for (int i = 0; i < PX; ++i) acc[0][i] = acc[1][i] = acc[2][i] = acc[3][i] = 0.0f;
#pragma unroll
                    for (int k = FL - 1; k >= 0; --k) {
                        const int sl = (u - k + 2 * FL) % FL;
                        const float t0 = a.taps[0][k], t1 = a.taps[1][k];
#pragma unroll
                        for (int i = 0; i < PX; ++i) {
                            acc[0][i] = fmaf(ring[0][sl][i], t0, acc[0][i]);
                            acc[1][i] = fmaf(ring[1][sl][i], t0, acc[1][i]);
                            acc[2][i] = fmaf(ring[0][sl][i], t1, acc[2][i]);
                            acc[3][i] = fmaf(ring[1][sl][i], t1, acc[3][i]);
                        }
                    }
                    float* o = a.out + (size_t)(v - (FL - 1)) * HW * 4;
#pragma unroll
                    for (int i = 0; i < PX; ++i)
                        if (ok[i])
                            *reinterpret_cast<float4*>(o + (size_t)px[i] * 4) = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
                }
            }
        }
    }
    if (bad && a.oob) atomicOr(a.oob, 1);
}

// Generic (any fl, any frame size) version: one thread per pixel per output frame, the window is re-read from
// the source (L2-served).  Used for fl > 32, for frame sizes that are not a multiple of 4 pixels and for still
// images (P == 2: out = (L_test, L_ref), fvvdp.py:251-253).
struct GenericArgs {
    const void* src[2];
    size_t chan_stride, frame_stride;
    int C, HW;
    EotfDev e;
    float w[3];
    int n_out, fl;
    float* out;
    int* oob;
    const float* taps;   // device [2][fl]
    const int* idx;      // device [fl-1+n_out]
};

template <int SRC, int P>
__global__ __launch_bounds__(256) void temporal_generic_kernel(const GenericArgs a) {
    __shared__ float lutw[SRC == SRC_U8 ? 768 : 1];
    if constexpr (SRC == SRC_U8) {
        build_lutw(lutw, a.e.lut, a.C, a.w, threadIdx.x, 256);
        __syncthreads();
    }
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    if (p >= a.HW) return;
    Sampler<SRC, 1> S[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        S[s].base = a.src[s];
        S[s].chan_stride = a.chan_stride;
        S[s].C = a.C;
        S[s].lutw = lutw;
        S[s].lut16 = a.e.lut;
        S[s].w0 = a.C == 3 ? a.w[0] : 1.0f;
        S[s].w1 = a.w[1];
        S[s].w2 = a.w[2];
        S[s].e = a.e;
    }
    bool bad = false;
    if constexpr (P == 2) {
        float lt[1], lr[1];
        const size_t off = (size_t)a.idx[t] * a.frame_stride + p;
        S[0].lum(off, lt, bad);
        S[1].lum(off, lr, bad);
        *reinterpret_cast<float2*>(a.out + ((size_t)t * a.HW + p) * 2) = make_float2(lt[0], lr[0]);
    } else {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = a.fl - 1; k >= 0; --k) {
            const size_t off = (size_t)a.idx[a.fl - 1 + t - k] * a.frame_stride + p;
            float lt[1], lr[1];
            S[0].lum(off, lt, bad);
            S[1].lum(off, lr, bad);
            const float f0 = a.taps[k], f1 = a.taps[a.fl + k];
            acc[0] = fmaf(lt[0], f0, acc[0]);
            acc[1] = fmaf(lr[0], f0, acc[1]);
            acc[2] = fmaf(lt[0], f1, acc[2]);
            acc[3] = fmaf(lr[0], f1, acc[3]);
        }
        *reinterpret_cast<float4*>(a.out + ((size_t)t * a.HW + p) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    if (bad && a.oob) atomicOr(a.oob, 1);
}

// planar [n][P][HW] <-> interleaved [n][HW][P]
template <int P>
__global__ void interleave_kernel(const float* __restrict__ in, float* __restrict__ out, int HW, int to_interleaved) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= HW) return;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        if (to_interleaved)
            out[((size_t)f * HW + p) * P + k] = in[((size_t)f * P + k) * HW + p];
        else
            out[((size_t)f * P + k) * HW + p] = in[((size_t)f * HW + p) * P + k];
    }
}

// ------------------------------------------------------------------------------------------------------------
// stage 2: fused pyramid level
//   read Gaussian level i once, write level i+1 once, and in the same pass expand level i+1, form the contrast
//   band, weight by the CSF, apply mutual masking and accumulate sum(D^beta)  (nothing else touches HBM).
//
//   One single-wave workgroup streams down a strip of 120 fine (60 coarse) columns: lane l owns coarse column
//   J = 60*strip+l and the two fine columns 2J, 2J+1 (4 fine pixels per step).  Vertical 5-tap reduce and the vertical
//   half of the expand are thread-local on a register window of 5 fine rows; the horizontal halves take the
//   neighbour lanes' values through DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- no LDS, no
//   barriers, so the waves of a CU run completely decoupled and hide each other's HBM latency.  Plane pairs
//   (test, ref) live in adjacent registers and go through packed fp32 math (v_pk_fma_f32), which also makes the
//   test and reference planes bit-symmetric (identical inputs give exactly D = 0).
// ------------------------------------------------------------------------------------------------------------
#define STRIP_J 60          // coarse columns produced per wave (64 lanes - 2 halo lanes each side)

typedef float v2f __attribute__((ext_vector_type(2)));

template <int P>
struct Px {                 // one pixel: P/2 (test, ref) pairs
    v2f h[P / 2];
};

template <int P>
__device__ __forceinline__ Px<P> ld_px(const float* p);
template <>
__device__ __forceinline__ Px<4> ld_px<4>(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    Px<4> r;
    r.h[0] = v2f{t.x, t.y};
    r.h[1] = v2f{t.z, t.w};
    return r;
}
template <>
__device__ __forceinline__ Px<2> ld_px<2>(const float* p) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    Px<2> r;
    r.h[0] = v2f{t.x, t.y};
    return r;
}
// the coarse level is written once and only read by the next launch: non-temporal stores (measured +2-3 % on the
// read+write mix of this kernel, tools/microbench/membw.hip)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_px(float* p, const Px<4>& a) {
    __builtin_nontemporal_store(v4f{a.h[0].x, a.h[0].y, a.h[1].x, a.h[1].y}, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ void st_px(float* p, const Px<2>& a) {
    __builtin_nontemporal_store(a.h[0], reinterpret_cast<v2f*>(p));
}

__device__ __forceinline__ v2f splat(float s) { return v2f{s, s}; }
// a*s + c on both halves (v_pk_fma_f32)
__device__ __forceinline__ v2f pfma(v2f a, float s, v2f c) { return __builtin_elementwise_fma(a, splat(s), c); }

// value held by the lane to the left / right (0 at the wave's ends)
__device__ __forceinline__ float from_left(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}
// acc + w * neighbour(x), per component (v_fmac_f32 with a DPP source)
__device__ __forceinline__ v2f fma_left(v2f x, float w, v2f acc) {
    return v2f{fmaf(from_left(x.x), w, acc.x), fmaf(from_left(x.y), w, acc.y)};
}
__device__ __forceinline__ v2f fma_right(v2f x, float w, v2f acc) {
    return v2f{fmaf(from_right(x.x), w, acc.x), fmaf(from_right(x.y), w, acc.y)};
}

struct BandArgs {
    const float* Gf;        // fine level   [n][h][w][P]
    float* Gc;              // coarse level [n][hc][wc][P]
    int w, h, wc, hc;
    int n_strips, n_chunks, cr;
    int n_items;            // work items (waves) of this launch
    int lut_lds;            // foveated: 1 = the band's LUT slice fits the dynamic LDS and is copied there
    float band_mul;
    const float4* csf;      // [32] records {S_log0[i], S_log1[i], S_log0[i+1]-S_log0[i], S_log1[i+1]-S_log1[i]}
    const float4* csf_y;    // [32] records {Y_log[i], ...} (foveated path: knots of the Y axis)
    float y_first, y_inv_step;
    float y_lo, y_hi;       // clamp range of L_bkg (lut Y[0], Y[-1])
    float ly_lo, ly_hi;     // the same in log2
    float lg_gain, lg_k;    // log2(sens_gain), log2(mask_k)
    float p, q0, q1, beta, lbkg_min, cmax, lg_dmax;
    float* partial;         // [n][n_strips*n_chunks][2]
    float* dD;
    float* dC;
    float* dL;
    float* dS;
    // foveated (FOV == true)
    const float4* sublut;   // per band: [32 ecc][32 Y][rw] of {S_log0[i], S_log1[i], S_log0[i+1], S_log1[i+1]} (i = rho knot)
    const float* axes;      // [3][32] knots: Y_log, rho_log, ecc_sqrt
    int rw, i_lo;           // rho knots covered by the band's sub-LUT: [i_lo, i_lo+rw]
    const float* fix;       // device [n][2]: gaze in frame pixels, or gaze view direction in degrees (map mode)
    const float* mvx;       // map mode (user geometry): view direction x,y [h][w] in degrees and resolution
    const float* mvy;       //   magnification [h][w] of this band, evaluated by the caller with the user's
    const float* mrm;       //   geometry object; nullptr = stock geometry computed in-kernel
    float size_m0, size_m1, dist_m, cos_delta, delta_rad;
    float rho_band, rho_lo, rho_hi, ecc_lo, ecc_hi;
    float inv_step[3], first[3];   // uniform-grid estimates of the three axes
    int frame_w, frame_h;
};


#define FOV_WPB 4            // foveated mode: 4 independent waves per workgroup share the band's LUT slice in LDS
extern __shared__ __attribute__((aligned(16))) float4 s_lut_dyn[];

template <int P, bool DBG, bool FOV>
__global__ __launch_bounds__(FOV ? 64 * FOV_WPB : 64, (FOV || DBG) ? 2 : 4) void band_kernel(const BandArgs a) {
    constexpr int HP = P / 2;   // (test, ref) pairs = temporal channels
    constexpr int WPB = FOV ? FOV_WPB : 1;
    __shared__ float4 s_csf[FVVDP_LUT_N];

    const int lane = FOV ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    // XCD-aware work order: hardware places workgroup b on XCD b % 8 (speed only, never correctness).  Give each
    // XCD a contiguous range of work items (strip fastest, then chunk, then frame) so that neighbouring strips,
    // which share their 4+4 halo columns, run on the same XCD at about the same time and hit in its L2.
    int bid;
    {
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
        bid = x * q8 + min(x, r8) + (blockIdx.x >> 3);
        if constexpr (FOV) bid = bid * WPB + (threadIdx.x >> 6);
    }
    const bool wave_has_work = !FOV || bid < a.n_items;
    const int strip = bid % a.n_strips;
    bid /= a.n_strips;
    const int chunk = bid % a.n_chunks;
    const int frame = bid / a.n_chunks;
    const int blk = chunk * a.n_strips + strip;

    const int w = a.w, h = a.h, wc = a.wc, hc = a.hc;
    // lane l of strip s owns coarse column J = 60*s + l, i.e. fine columns 120*s + 2l (+1): the 128 fine pixels a
    // wave reads per row start at byte 1920*s of the row -> aligned to the 128-byte lines.  Lanes 2..61 produce
    // output; lanes 0,1 and 62,63 only feed their neighbours (in strip 0 the image border makes lanes 0,1 complete).
    const int J = strip * STRIP_J + lane;
    const int ca = chunk * a.cr;
    const int cb = min(ca + a.cr, hc);
    const bool active = (lane >= 2 || strip == 0) && (lane < 62) && (J < wc);
    const int X0 = 2 * J, X1 = 2 * J + 1;
    const int xc0 = min(max(X0, 0), w - 1), xc1 = min(max(X1, 0), w - 1);
    const bool col1_ok = X1 < w;

    __shared__ float2 s_ax[FOV ? 3 * FVVDP_LUT_N : 1];     // {knot k, 1/(knot k+1 - knot k + 1e-6)} of the three axes
    if constexpr (!FOV) {
        if (lane < FVVDP_LUT_N) s_csf[lane] = a.csf[lane];
    } else {
        for (int i = threadIdx.x; i < 3 * FVVDP_LUT_N; i += 64 * WPB) {
            const int k = i % FVVDP_LUT_N;
            const float x0 = a.axes[i];
            const float x1 = a.axes[k + 1 < FVVDP_LUT_N ? i + 1 : i];
            s_ax[i] = make_float2(x0, 1.0f / (x1 - x0 + 0.000001f));
        }
        if (a.lut_lds) {
            const int nl = FVVDP_LUT_N * FVVDP_LUT_N * a.rw;
            for (int i = threadIdx.x; i < nl; i += 64 * WPB) s_lut_dyn[i] = a.sublut[i];
        }
    }
    __syncthreads();
    if constexpr (FOV) {
        if (!wave_has_work) return;
    }

    // horizontal 5-tap weights of this lane's coarse column incl. the reference's edge fix-ups
    // (gausspyr_reduce, fvvdp_lpyr_dec.py:198-205; the right-edge branch is selected by the parity of the ROW
    // count, :202, reproduced here on purpose).  Taps: E[l-1], O[l-1], E[l], O[l], E[l+1]  (E/O = even/odd fine
    // column of a lane); taps falling outside the image get weight 0 and their fix-up is folded into the
    // in-range taps.
    const float K0 = 0.05f, K1 = 0.25f, K2 = 0.4f, K3 = 0.25f, K4 = 0.05f;
    float wq0 = K0, wq1 = K1, wq2 = K2, wq3 = K3, wq4 = K4;
    if (J == 0) {
        wq2 += K1;
        wq3 += K0;
        wq0 = 0.0f;
        wq1 = 0.0f;
    }
    if (J == wc - 1) {
        const bool hodd = (h & 1) != 0;
        if (w & 1) {   // own columns: X0 = w-1 (tap 2), X1 = w (outside)
            wq3 = 0.0f;
            wq4 = 0.0f;
            if (hodd) { wq2 += K3; wq1 += K4; } else { wq2 += K4; }
        } else {       // own columns: w-2 (tap 2), w-1 (tap 3); tap 4 = column w is outside
            wq4 = 0.0f;
            if (hodd) { wq3 += K3; wq2 += K4; } else { wq3 += K4; }
        }
    }
    // horizontal expand weights (2K = .1 .8 .1 / .5 .5, gausspyr_expand fvvdp_lpyr_dec.py:126-142,233); a
    // neighbour outside the coarse row is the clamped (own) column, so its weight moves to the centre tap
    const bool at_l = (J <= 0), at_r = (J >= wc - 1);
    const float el = at_l ? 0.0f : 0.1f, er = at_r ? 0.0f : 0.1f;
    const float ec = 0.8f + (at_l ? 0.1f : 0.0f) + (at_r ? 0.1f : 0.0f);
    const float orr = at_r ? 0.0f : 0.5f;
    const float oc = at_r ? 1.0f : 0.5f;

    const float* Gf = a.Gf + (size_t)frame * h * w * P;
    float* Gc = a.Gc + (size_t)frame * hc * wc * P;

    auto load_row = [&](int r, Px<P>& p0, Px<P>& p1) {
        int rr = r < 0 ? -1 - r : (r >= h ? 2 * h - 1 - r : r);   // symmetric padding (fvvdp_lpyr_dec.py:190-195)
        rr = min(max(rr, 0), h - 1);
        const float* row = Gf + (size_t)rr * w * P;
        p0 = ld_px<P>(row + (size_t)xc0 * P);
        p1 = ld_px<P>(row + (size_t)xc1 * P);
    };

    Px<P> W[5][2];

    // one coarse row from the current window: vertical 5-tap in registers, horizontal 5-tap across lanes
    auto coarse_step = [&]() -> Px<P> {
        Px<P> c, va, vb;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f a0 = W[0][0].h[k] * K0;
            a0 = pfma(W[1][0].h[k], K1, a0);
            a0 = pfma(W[2][0].h[k], K2, a0);
            a0 = pfma(W[3][0].h[k], K3, a0);
            va.h[k] = pfma(W[4][0].h[k], K4, a0);
            v2f b0 = W[0][1].h[k] * K0;
            b0 = pfma(W[1][1].h[k], K1, b0);
            b0 = pfma(W[2][1].h[k], K2, b0);
            b0 = pfma(W[3][1].h[k], K3, b0);
            vb.h[k] = pfma(W[4][1].h[k], K4, b0);
        }
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f acc = va.h[k] * wq2;
            acc = pfma(vb.h[k], wq3, acc);
            acc = fma_left(va.h[k], wq0, acc);
            acc = fma_left(vb.h[k], wq1, acc);
            c.h[k] = fma_right(va.h[k], wq4, acc);
        }
        return c;
    };
    auto shift_window = [&](const Px<P> (&n0)[2], const Px<P> (&n1)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            W[0][j] = W[2][j];
            W[1][j] = W[3][j];
            W[2][j] = W[4][j];
            W[3][j] = n0[j];
            W[4][j] = n1[j];
        }
    };

    // ---- prologue: coarse rows ca-1 and ca --------------------------------------------------------------
    {
        const int r0 = 2 * (ca - 1) - 2;
#pragma unroll
        for (int k = 0; k < 5; ++k) load_row(r0 + k, W[k][0], W[k][1]);
    }
    const Px<P> cA = coarse_step();
    Px<P> nx0[2], nx1[2];
    load_row(2 * ca + 1, nx0[0], nx0[1]);
    load_row(2 * ca + 2, nx1[0], nx1[1]);
    shift_window(nx0, nx1);
    const Px<P> cB = coarse_step();
    if (active) st_px(Gc + ((size_t)ca * wc + J) * P, cB);
    Px<P> Gm1 = (ca > 0) ? cA : cB;
    Px<P> G0 = cB;
    load_row(2 * ca + 3, nx0[0], nx0[1]);
    load_row(2 * ca + 4, nx1[0], nx1[1]);

    float acc[2] = {0.0f, 0.0f};

    // foveated constants of this lane's two fine columns
    float vxa = 0.0f, vxb = 0.0f, gx = 0.0f, gy = 0.0f;
    if constexpr (FOV) {
        // pix2view_direction (fvvdp_display_model.py:498-510) on the band grid, pixel centres at +0.5
        const float xa = ((float)X0 + 0.5f) + (-(float)w / 2.0f);
        const float xb = ((float)X1 + 0.5f) + (-(float)w / 2.0f);
        vxa = atanf(xa * a.size_m0 / (float)w / a.dist_m) * 57.29577951308232f;
        vxb = atanf(xb * a.size_m0 / (float)w / a.dist_m) * 57.29577951308232f;
        if (a.mvx) {
            gx = a.fix[2 * frame + 0];
            gy = a.fix[2 * frame + 1];
        } else {
            const float fxp = a.fix[2 * frame + 0] + 0.5f, fyp = a.fix[2 * frame + 1] + 0.5f;
            const float gxm = (fxp + (-(float)a.frame_w / 2.0f)) * a.size_m0 / (float)a.frame_w;
            const float gym = -(fyp + (-(float)a.frame_h / 2.0f)) * a.size_m1 / (float)a.frame_h;
            gx = atanf(gxm / a.dist_m) * 57.29577951308232f;
            gy = atanf(gym / a.dist_m) * 57.29577951308232f;
        }
    }

    const float lg_bm = __log2f(a.band_mul);
    const float lg_base = a.lg_gain;              // log2(S) = interp + log2(gain)      (fvvdp.py:447)
    const float lg_mask = a.lg_gain + a.lg_k;      // log2(k*S)

    // per-pixel tail: contrast, CSF, masking, pooling  (fvvdp_lpyr_dec.py:259-269, fvvdp.py:395-467)
    auto band_px = [&](const Px<P>& g, const Px<P>& e, bool valid, int y, int x, float vx, float vy, float res_mag) {
        (void)x; (void)res_mag;
        const float lb = fmaxf(e.h[0].y, a.lbkg_min);                  // plane 1 = reference (sustained)
        // contrast = min((g-e)/lb, cmax) * m.  Dividing by lb>0 commutes with |.|, min and the clamp, so the
        // division is carried as -log2(lb) in the log domain below: no reciprocal, no per-plane multiply.
        const float dcap = a.cmax * lb;                                // (g-e)/lb <= cmax  <=>  g-e <= cmax*lb
        v2f d[HP];
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            const v2f t = g.h[k] - e.h[k];
            d[k] = v2f{fminf(t.x, dcap), fminf(t.y, dcap)};            // upper clamp only (fvvdp_lpyr_dec.py:266)
        }
        const float llb = fast_log2(lb);
        const float yq = fminf(fmaxf(llb, a.ly_lo), a.ly_hi);          // = log2(clamp(lb, Y[0], Y[-1]))  (fvvdp.py:530)
        float slog[2] = {0.0f, 0.0f};
        if constexpr (!FOV) {
            // 1-D table over log2(L_bkg) (uniform knots): interval from the grid, value = v[i] + f*(v[i+1]-v[i])
            const float t = (yq - a.y_first) * a.y_inv_step;
            const float fi = fminf(fmaxf(floorf(t), 0.0f), (float)(FVVDP_LUT_N - 2));
            const float4 r = s_csf[(int)fi];                           // {v0[i], v1[i], v0[i+1]-v0[i], v1[i+1]-v1[i]}
            const float f = t - fi;
            slog[0] = fmaf(f, r.z, r.x);
            slog[1] = fmaf(f, r.w, r.y);
        } else {
            // eccentricity and resolution magnification (fvvdp.py:424-437, fvvdp_display_model.py:475-526).
            // (tan(a+d)-tan(a))/tan(d) == cos(d)/(cos(a)cos(a+d)): evaluated in this form it needs no slow tan and
            // does not lose digits to the reference's fp32 finite difference (whose noise, ~5e-4, bounds parity).
            const float dx = vx - gx, dy = vy - gy;
            const float ecc = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
            const float rho = a.rho_band * res_mag;
            const float rq = fast_log2(fminf(fmaxf(rho, a.rho_lo), a.rho_hi));
            const float eq = __builtin_amdgcn_sqrtf(fminf(fmaxf(ecc, a.ecc_lo), a.ecc_hi));
            // interval on each (uniform) axis from the grid, fraction from the stored knots incl. interp.py:16's +1e-6
            auto axis = [&](int ax, float q, int lo, int hi, int& k, float& f) {
                k = min(max((int)floorf((q - a.first[ax]) * a.inv_step[ax]), lo), hi);
                const float2 kn = s_ax[ax * FVVDP_LUT_N + k];
                f = fmaxf((q - kn.x) * kn.y, 0.0f);
            };
            int kY, kR, kE;
            float fY, fR, fE;
            axis(0, yq, 0, FVVDP_LUT_N - 2, kY, fY);
            axis(1, rq, a.i_lo, a.i_lo + a.rw - 1, kR, fR);
            axis(2, eq, 0, FVVDP_LUT_N - 2, kE, fE);
            const int so = (kE * FVVDP_LUT_N + kY) * a.rw + (kR - a.i_lo);
            const int sj = a.rw, sk = FVVDP_LUT_N * a.rw;
            float4 v00, v10, v01, v11;                                                      // v[dj][dk]
            if (a.lut_lds) {
                v00 = s_lut_dyn[so]; v10 = s_lut_dyn[so + sj]; v01 = s_lut_dyn[so + sk]; v11 = s_lut_dyn[so + sk + sj];
            } else {
                const float4* sb = a.sublut + so;
                v00 = sb[0]; v10 = sb[sj]; v01 = sb[sk]; v11 = sb[sk + sj];
            }
            const float gR = 1.0f - fR, gY = 1.0f - fY, gE = 1.0f - fE;
            // interp3 (interp.py:53-57), same association: rho blend, then Y, then ecc
            slog[0] = ((v00.x * gR + v00.z * fR) * gY + (v10.x * gR + v10.z * fR) * fY) * gE +
                      ((v01.x * gR + v01.z * fR) * gY + (v11.x * gR + v11.z * fR) * fY) * fE;
            slog[1] = ((v00.y * gR + v00.w * fR) * gY + (v10.y * gR + v10.w * fR) * fY) * gE +
                      ((v01.y * gR + v01.w * fR) * gY + (v11.y * gR + v11.w * fR) * fY) * fE;
        }
        const float vm = valid ? 1.0f : 0.0f;
        const float lcn = lg_bm - llb;                                   // log2(m / lb)
        // D = |T'-R'|^p / (1 + (k*min(|T'|,|R'|))^q), T' = T*S   (fvvdp.py:585-595), in the log2 domain.
        // Video: the two temporal channels are carried as one (sustained, transient) pair through packed fp32 ops;
        // only the transcendentals are per component.
        float ldd_dbg[2] = {0.0f, 0.0f};
        if constexpr (HP == 2) {
            const v2f sl = v2f{slog[0], slog[1]};
            const v2f lsb = sl + splat(lcn + lg_base);                   // log2(S * m / lb)
            const v2f lsm = sl + splat(lcn + lg_mask);                   // log2(k * S * m / lb)
            const v2f ldiff = v2f{fast_log2(fabsf(d[0].x - d[0].y)), fast_log2(fabsf(d[1].x - d[1].y))};
            const v2f lmin = v2f{fast_log2(fminf(fabsf(d[0].x), fabsf(d[0].y))), fast_log2(fminf(fabsf(d[1].x), fabsf(d[1].y)))};
            const v2f ld = (ldiff + lsb) * splat(a.p);
            const v2f lm = (lmin + lsm) * v2f{a.q0, a.q1};
            const v2f one_mq = v2f{fast_exp2(lm.x), fast_exp2(lm.y)} + splat(1.0f);
            const v2f t = ld - v2f{fast_log2(one_mq.x), fast_log2(one_mq.y)};
            const v2f ldd = v2f{fminf(t.x, a.lg_dmax), fminf(t.y, a.lg_dmax)};
            const v2f bl = ldd * splat(a.beta);
            const v2f term = v2f{fast_exp2(bl.x), fast_exp2(bl.y)};      // D^beta for the spatial pooling (fvvdp.py:467,607)
            const v2f av = __builtin_elementwise_fma(term, splat(vm), v2f{acc[0], acc[1]});
            acc[0] = av.x;
            acc[1] = av.y;
            ldd_dbg[0] = ldd.x;
            ldd_dbg[1] = ldd.y;
        } else {
            const float dT = d[0].x, dR = d[0].y;
            const float ls = slog[0] + lcn;
            const float ld = a.p * (fast_log2(fabsf(dT - dR)) + (ls + lg_base));
            const float mq = fast_exp2(a.q0 * (fast_log2(fminf(fabsf(dT), fabsf(dR))) + (ls + lg_mask)));
            const float ldd = fminf(ld - fast_log2(1.0f + mq), a.lg_dmax);
            acc[0] = fmaf(fast_exp2(a.beta * ldd), vm, acc[0]);
            ldd_dbg[0] = ldd;
        }
        if constexpr (DBG) {
            if (valid) {
#pragma unroll
                for (int cc = 0; cc < HP; ++cc) {
                    const size_t o = (((size_t)frame * 2 + cc) * h + y) * w + x;
                    if (a.dD) a.dD[o] = fast_exp2(ldd_dbg[cc]);
                    if (a.dS) a.dS[o] = fast_exp2(slog[cc]);
                }
            }
        }
        if constexpr (DBG) {
            if (valid) {
                if (a.dC) {
                    const float sc = a.band_mul / lb;
#pragma unroll
                    for (int k = 0; k < HP; ++k) {
                        a.dC[(((size_t)frame * P + 2 * k) * h + y) * w + x] = d[k].x * sc;
                        a.dC[(((size_t)frame * P + 2 * k + 1) * h + y) * w + x] = d[k].y * sc;
                    }
                }
                if (a.dL) a.dL[((size_t)frame * h + y) * w + x] = lb;
            }
        }
    };

    // ---- main loop: band rows 2c, 2c+1 for c in [ca, cb) ------------------------------------------------
    for (int c = ca; c < cb; ++c) {
        shift_window(nx0, nx1);               // window = fine rows 2c .. 2c+4
        if (c + 1 < cb) {                     // prefetch the two rows of the next step
            load_row(2 * c + 5, nx0[0], nx0[1]);
            load_row(2 * c + 6, nx1[0], nx1[1]);
        }
        const Px<P> cN = coarse_step();       // coarse row c+1
        const bool has_next = (c + 1) <= (hc - 1);
        Px<P> Gp1 = has_next ? cN : G0;       // index clamp of the expand (fvvdp_lpyr_dec.py:134,138)
        if (has_next && (c + 1) < cb && active) st_px(Gc + ((size_t)(c + 1) * wc + J) * P, cN);
        Px<P> x00, x01, x10, x11;             // expanded level at (row 2c|2c+1, col X0|X1)
        Px<P> evE, evO;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            // vertical expand on the coarse column: even fine row 2c (.1 .8 .1), odd fine row 2c+1 (.5 .5)
            v2f t = Gm1.h[k] * 0.1f;
            t = pfma(G0.h[k], 0.8f, t);
            evE.h[k] = pfma(Gp1.h[k], 0.1f, t);
            evO.h[k] = pfma(Gp1.h[k], 0.5f, G0.h[k] * 0.5f);
        }
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f t = evE.h[k] * ec;
            t = fma_left(evE.h[k], el, t);
            x00.h[k] = fma_right(evE.h[k], er, t);
            x01.h[k] = fma_right(evE.h[k], orr, evE.h[k] * oc);
            t = evO.h[k] * ec;
            t = fma_left(evO.h[k], el, t);
            x10.h[k] = fma_right(evO.h[k], er, t);
            x11.h[k] = fma_right(evO.h[k], orr, evO.h[k] * oc);
        }
        const bool row1_ok = (2 * c + 1) < h;
#if defined(BAND_ABLATE) && BAND_ABLATE >= 1      // profiling ablation: no per-pixel tail, keep the data flow alive
        acc[0] += x00.h[0].x + x01.h[0].x + x10.h[0].x + x11.h[0].x + W[0][0].h[0].x + W[0][1].h[0].x + W[1][0].h[0].x + W[1][1].h[0].x;
        if (false)
#endif
        {
        float vy0 = 0.0f, vy1 = 0.0f;        // vertical view angle of the two fine rows (foveated)
        if constexpr (FOV) {
            const float yp0 = ((float)(2 * c) + 0.5f) + (-(float)h / 2.0f), yp1 = yp0 + 1.0f;
            vy0 = atanf(-yp0 * a.size_m1 / (float)h / a.dist_m) * 57.29577951308232f;
            vy1 = atanf(-yp1 * a.size_m1 / (float)h / a.dist_m) * 57.29577951308232f;
        }
        if constexpr (FOV) {
            float vx4[4] = {vxa, vxb, vxa, vxb}, vy4[4] = {vy0, vy0, vy1, vy1}, rm4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (a.mvx) {                          // user geometry: maps evaluated by the caller
                const int ya = min(2 * c, h - 1), yb = min(2 * c + 1, h - 1);
                const size_t o[4] = {(size_t)ya * w + xc0, (size_t)ya * w + xc1, (size_t)yb * w + xc0, (size_t)yb * w + xc1};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vx4[i] = a.mvx[o[i]];
                    vy4[i] = a.mvy[o[i]];
                    rm4[i] = a.mrm[o[i]];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float va = fminf(__builtin_amdgcn_sqrtf(vx4[i] * vx4[i] + vy4[i] * vy4[i]), 89.9f) * 0.017453292519943295f;
                    rm4[i] = a.cos_delta * fast_rcp(__cosf(va) * __cosf(va + a.delta_rad));
                }
            }
            band_px(W[0][0], x00, active, 2 * c, X0, vx4[0], vy4[0], rm4[0]);
            band_px(W[0][1], x01, active && col1_ok, 2 * c, X1, vx4[1], vy4[1], rm4[1]);
            band_px(W[1][0], x10, active && row1_ok, 2 * c + 1, X0, vx4[2], vy4[2], rm4[2]);
            band_px(W[1][1], x11, active && row1_ok && col1_ok, 2 * c + 1, X1, vx4[3], vy4[3], rm4[3]);
        } else {
            band_px(W[0][0], x00, active, 2 * c, X0, 0.0f, 0.0f, 1.0f);
            band_px(W[0][1], x01, active && col1_ok, 2 * c, X1, 0.0f, 0.0f, 1.0f);
            band_px(W[1][0], x10, active && row1_ok, 2 * c + 1, X0, 0.0f, 0.0f, 1.0f);
            band_px(W[1][1], x11, active && row1_ok && col1_ok, 2 * c + 1, X1, 0.0f, 0.0f, 1.0f);
        }
        }
        Gm1 = G0;
        G0 = Gp1;
    }

    const float s0 = wave_sum(acc[0]);
    const float s1 = wave_sum(acc[1]);
    if (lane == 0) {
        float* o = a.partial + ((size_t)frame * (a.n_strips * a.n_chunks) + blk) * 2;
        o[0] = s0;
        o[1] = s1;
    }
}

// Q[band][cc][slot] = (sum D^beta / n_px)^(1/beta)   (lp_norm, fvvdp.py:598-607); fixed summation order.
struct FinalizeArgs {
    const float* partial;
    float* Q;
    int n_bands, n, q_stride, q_col0, tc;
    float inv_beta;
    int nblk[FVVDP_MAX_BANDS];
    long long off[FVVDP_MAX_BANDS];
    float npx[FVVDP_MAX_BANDS];
};

__global__ __launch_bounds__(64) void finalize_kernel(const FinalizeArgs a) {
    // one wave per (band, cc, slot); lane l adds partials l, l+64, ... in fp64, then a fixed shuffle tree
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const int s = i % a.n;
    const int cc = (i / a.n) % 2;
    const int b = i / (2 * a.n);
    float q = 0.0f;
    if (cc < a.tc) {
        const float* p = a.partial + a.off[b] + (size_t)s * a.nblk[b] * 2 + cc;
        double sum = 0.0;
        for (int k = lane; k < a.nblk[b]; k += 64) sum += (double)p[2 * k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
        q = (float)pow(sum / (double)a.npx[b], (double)a.inv_beta);
    }
    if (lane == 0) a.Q[((size_t)b * 2 + cc) * a.q_stride + a.q_col0 + s] = q;
}

// Heat-map reconstruction, one level: out = expand(coarse) + (D0 + w*D1)/m   [then ^beta_jod * |jod_a| on level 0]
// (heatmap_pyr.set_band / reconstruct, fvvdp_lpyr_dec.py:65-71,94-101; expand closed form as in band_kernel).
struct HeatArgs {
    const float* D;        // [n][2][h][w]
    const float* coarse;   // [n][hc][wc] or nullptr for the coarsest band
    float* out;            // [n][h][w]
    int w, h, wc, hc, tc;
    float w_trans, inv_m, beta_jod, scale;
    int final_level;
};

__global__ __launch_bounds__(256) void heat_level_kernel(const HeatArgs a) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (x >= a.w) return;
    const size_t plane = (size_t)a.h * a.w;
    const size_t o = (size_t)y * a.w + x;
    float v = a.D[((size_t)f * 2) * plane + o];
    if (a.tc == 2) v = v + a.w_trans * a.D[((size_t)f * 2 + 1) * plane + o];
    v = v * a.inv_m;
    if (a.coarse) {
        const float* c = a.coarse + (size_t)f * a.hc * a.wc;
        const int cy = y >> 1, cx = x >> 1;
        const int r0 = max(cy - 1, 0), r1 = cy, r2 = min(cy + 1, a.hc - 1);
        const int c0 = max(cx - 1, 0), c1 = cx, c2 = min(cx + 1, a.wc - 1);
        auto col = [&](int cc) -> float {     // vertical pass first (gausspyr_expand, fvvdp_lpyr_dec.py:225-228)
            if (y & 1) return 0.5f * c[(size_t)r1 * a.wc + cc] + 0.5f * c[(size_t)r2 * a.wc + cc];
            return (0.1f * c[(size_t)r0 * a.wc + cc] + 0.8f * c[(size_t)r1 * a.wc + cc]) + 0.1f * c[(size_t)r2 * a.wc + cc];
        };
        float e;
        if (x & 1) e = 0.5f * col(c1) + 0.5f * col(c2);
        else e = (0.1f * col(c0) + 0.8f * col(c1)) + 0.1f * col(c2);
        v = e + v;
    }
    if (a.final_level) v = powf(v, a.beta_jod) * a.scale;
    a.out[(size_t)f * plane + o] = v;
}

// ------------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------------
struct fvvdp_ctx {
    int W = 0, H = 0, n_bands = 0, P = 0, max_frames = 0;
    fvvdp_params prm{};
    double rho_band[FVVDP_MAX_BANDS + 1]{};
    int lw[FVVDP_MAX_BANDS + 1]{}, lh[FVVDP_MAX_BANDS + 1]{};
    float* level[FVVDP_MAX_BANDS + 1]{};
    float* partial = nullptr;
    long long partial_off[FVVDP_MAX_BANDS]{};
    int max_blk[FVVDP_MAX_BANDS]{};
    size_t partial_floats = 0;
    float4* csf = nullptr;        // [n_bands][32] slope-form records of the 1-D tables
    float4* csf_y = nullptr;      // [32] {Y_log[i],0,0,0}
    bool csf_set = false;
    float y_first = 0, y_inv_step = 0, y_lo = 0, y_hi = 0;
    std::vector<float> h_lut3[2];              // host copies of the full 32^3 LUTs (foveated mode)
    float h_axes[3][FVVDP_LUT_N]{};            // Y_log, rho_log, ecc_sqrt
    float* d_axes = nullptr;                   // [3][32]
    float4* sublut[FVVDP_MAX_BANDS]{};         // per-band rho slices, rebuilt when the geometry changes
    int sub_rw[FVVDP_MAX_BANDS]{}, sub_ilo[FVVDP_MAX_BANDS]{};
    fvvdp_geom sub_geom{};
    const float* map_vx[FVVDP_MAX_BANDS]{};    // user-geometry maps (owned by the caller)
    const float* map_vy[FVVDP_MAX_BANDS]{};
    const float* map_rm[FVVDP_MAX_BANDS]{};
    float map_rm_max[FVVDP_MAX_BANDS]{}, map_rm_min[FVVDP_MAX_BANDS]{};
    bool maps_set = false;
    bool sub_valid = false;
    float rho_lo = 0, rho_hi = 0, ecc_lo = 0, ecc_hi = 0;
    bool lut3_set[2] = {false, false};
    float* d_fix = nullptr;       // [max_frames][2]
    float* d_taps = nullptr;      // [2][FVVDP_MAX_TAPS]
    int* d_idx = nullptr;         // [max_frames + FVVDP_MAX_TAPS]
    float* heat[FVVDP_MAX_BANDS + 1]{};   // heat-map accumulation images of levels >= 1, allocated on first use
    size_t scratch = 0;
    long long wave_capacity = 4096;   // resident single-wave workgroups of the band kernel on the whole chip
    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[FVVDP_MAX_BANDS + 2];
    float t_ms[FVVDP_MAX_BANDS + 2]{};
    int t_cnt[FVVDP_MAX_BANDS + 2]{};
};

template <typename T>
static int dev_alloc(fvvdp_ctx* c, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T));
    if (e != hipSuccess) return fail(FVVDP_ENOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    *p = reinterpret_cast<T*>(q);
    c->scratch += count * sizeof(T);
    return FVVDP_OK;
}

struct Timed {
    fvvdp_ctx* c;
    int id;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    Timed(fvvdp_ctx* c_, int id_, hipStream_t st_) : c(c_), id(id_), st(st_) {
        if (c->timing) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, st);
        }
    }
    ~Timed() {
        if (c->timing) {
            (void)hipEventRecord(e1, st);
            c->ev[id].push_back({e0, e1});
        }
    }
};

// strips cover coarse columns [0,62), [62,122), ... (see band_kernel)
static int band_strips(int wc) { return wc <= 62 ? 1 : 1 + (wc - 62 + STRIP_J - 1) / STRIP_J; }

static void chunking(int hc, int n_strips, int n, long long capacity, int& n_chunks, int& cr) {
    // Every single-wave workgroup does the same amount of work (cr steps + the prologue for the two halo coarse
    // rows, whose 7 extra fine rows are re-read from HBM: weighted as 8 steps), so the launch proceeds in "rounds"
    // of `capacity` resident waves.  Pick the chunk height that minimises rounds x per-wave cost.
    double best = 1e300;
    cr = hc;
    for (int cand = 2; cand <= hc || cand == 2; ++cand) {
        const int c = cand > hc ? hc : cand;
        const long long chunks = (hc + c - 1) / c;
        const long long waves = (long long)n * n_strips * chunks;
        const long long rounds = (waves + capacity - 1) / capacity;
        const double cost = (double)rounds * ((double)c + 8.0) * (1.0 + 1e-4 * (double)chunks);
        if (cost < best) { best = cost; cr = c; }
        if (cand >= hc) break;
    }
    if (const char* ov = getenv("FVVDP_BAND_CR")) {       // tuning override
        const int v = atoi(ov);
        if (v >= 1) cr = v > hc ? hc : v;
    }
    n_chunks = (hc + cr - 1) / cr;
}

extern "C" int fvvdp_ctx_create(fvvdp_ctx** out, int width, int height, int n_bands, int planes, int max_frames,
                                const double* h_rho_band, const fvvdp_params* prm) {
    if (!out || !prm || !h_rho_band) return fail(FVVDP_EINVAL, "null argument");
    if (planes != 2 && planes != 4) return fail(FVVDP_EINVAL, "planes must be 2 (image) or 4 (video), got %d", planes);
    if (n_bands < 1 || n_bands > FVVDP_MAX_BANDS) return fail(FVVDP_EINVAL, "n_bands %d out of range", n_bands);
    if (max_frames < 1) return fail(FVVDP_EINVAL, "max_frames must be >= 1");
    if (width < 4 || height < 4) return fail(FVVDP_EINVAL, "frame %dx%d too small", width, height);
    fvvdp_ctx* c = new fvvdp_ctx();
    c->W = width;
    c->H = height;
    c->n_bands = n_bands;
    c->P = planes;
    c->max_frames = max_frames;
    c->prm = *prm;
    int w = width, h = height;
    for (int i = 0; i <= n_bands; ++i) {
        c->lw[i] = w;
        c->lh[i] = h;
        c->rho_band[i] = h_rho_band[i];
        if (i < n_bands && (w < 2 || h < 2)) {
            delete c;
            return fail(FVVDP_EINVAL, "pyramid level %d is %dx%d: too many bands for this frame size", i, w, h);
        }
        w = (w + 1) / 2;
        h = (h + 1) / 2;
    }
    int rc = FVVDP_OK;
    for (int i = 0; i <= n_bands && rc == FVVDP_OK; ++i)
        rc = dev_alloc(c, &c->level[i], (size_t)max_frames * c->lw[i] * c->lh[i] * planes);
    size_t off = 0;
    for (int b = 0; b < n_bands; ++b) {
        const int n_strips = band_strips(c->lw[b + 1]);
        const int max_chunks = (c->lh[b + 1] + 1) / 2;
        c->max_blk[b] = n_strips * max_chunks;
        c->partial_off[b] = (long long)off;
        off += (size_t)max_frames * c->max_blk[b] * 2;
    }
    c->partial_floats = off;
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->partial, off);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->csf, (size_t)n_bands * FVVDP_LUT_N);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->csf_y, (size_t)FVVDP_LUT_N);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->d_fix, (size_t)max_frames * 2);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->d_taps, (size_t)2 * FVVDP_MAX_TAPS);
    if (rc == FVVDP_OK) rc = dev_alloc(c, &c->d_idx, (size_t)max_frames + FVVDP_MAX_TAPS);
    if (rc != FVVDP_OK) {
        fvvdp_ctx_destroy(c);
        return rc;
    }
    {
        int dev = 0, cus = 256, per_cu = 16;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        }
        hipError_t e = (planes == 4)
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, band_kernel<4, false, false>, 64, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, band_kernel<2, false, false>, 64, 0);
        if (e != hipSuccess || per_cu < 1) per_cu = 16;
        c->wave_capacity = (long long)per_cu * cus;
    }
    *out = c;
    return FVVDP_OK;
}

extern "C" void fvvdp_ctx_destroy(fvvdp_ctx* c) {
    if (!c) return;
    for (int i = 0; i <= FVVDP_MAX_BANDS; ++i)
        if (c->level[i]) (void)hipFree(c->level[i]);
    if (c->partial) (void)hipFree(c->partial);
    if (c->csf) (void)hipFree(c->csf);
    if (c->csf_y) (void)hipFree(c->csf_y);
    for (int i = 0; i <= FVVDP_MAX_BANDS; ++i)
        if (c->heat[i]) (void)hipFree(c->heat[i]);
    if (c->d_fix) (void)hipFree(c->d_fix);
    if (c->d_taps) (void)hipFree(c->d_taps);
    if (c->d_idx) (void)hipFree(c->d_idx);
    if (c->d_axes) (void)hipFree(c->d_axes);
    for (int b = 0; b < FVVDP_MAX_BANDS; ++b)
        if (c->sublut[b]) (void)hipFree(c->sublut[b]);
    for (auto& v : c->ev)
        for (auto& pr : v) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    delete c;
}

extern "C" int fvvdp_ctx_level_size(const fvvdp_ctx* c, int level, int* w, int* h) {
    if (!c || level < 0 || level > c->n_bands) return fail(FVVDP_EINVAL, "bad level %d", level);
    if (w) *w = c->lw[level];
    if (h) *h = c->lh[level];
    return FVVDP_OK;
}

extern "C" size_t fvvdp_ctx_scratch_bytes(const fvvdp_ctx* c) { return c ? c->scratch : 0; }

extern "C" int fvvdp_ctx_set_csf_1d(fvvdp_ctx* c, const float* h_Y_log, const float* h_S_log) {
    if (!c || !h_Y_log || !h_S_log) return fail(FVVDP_EINVAL, "null argument");
    // Record i of a band holds the table value at knot i and the step to knot i+1 for both temporal channels:
    // the kernel evaluates v[i] + f*(v[i+1]-v[i]) with f from the (uniform) knot grid.
    std::vector<float4> rec((size_t)c->n_bands * FVVDP_LUT_N);
    for (int b = 0; b < c->n_bands; ++b)
        for (int i = 0; i < FVVDP_LUT_N; ++i) {
            const float* v0 = h_S_log + ((size_t)b * 2 + 0) * FVVDP_LUT_N;
            const float* v1 = h_S_log + ((size_t)b * 2 + 1) * FVVDP_LUT_N;
            const int j = i + 1 < FVVDP_LUT_N ? i + 1 : i;
            rec[(size_t)b * FVVDP_LUT_N + i] = make_float4(v0[i], v1[i], v0[j] - v0[i], v1[j] - v1[i]);
        }
    HIP_TRY(hipMemcpy(c->csf, rec.data(), rec.size() * sizeof(float4), hipMemcpyHostToDevice));
    c->y_first = h_Y_log[0];
    c->y_inv_step = (float)(FVVDP_LUT_N - 1) / (h_Y_log[FVVDP_LUT_N - 1] - h_Y_log[0]);
    c->y_lo = exp2f(h_Y_log[0]);
    c->y_hi = exp2f(h_Y_log[FVVDP_LUT_N - 1]);
    c->csf_set = true;
    return FVVDP_OK;
}

extern "C" int fvvdp_ctx_set_csf_3d(fvvdp_ctx* c, int tc, const float* h_S_log, const float* h_Y_log,
                                    const float* h_rho_log, const float* h_ecc_sqrt) {
    if (!c || !h_S_log || !h_Y_log || !h_rho_log || !h_ecc_sqrt) return fail(FVVDP_EINVAL, "null argument");
    if (tc < 0 || tc > 1) return fail(FVVDP_EINVAL, "temporal_channel must be 0 or 1");
    const size_t n3 = (size_t)FVVDP_LUT_N * FVVDP_LUT_N * FVVDP_LUT_N;
    int rc = FVVDP_OK;
    if (!c->d_axes) rc = dev_alloc(c, &c->d_axes, (size_t)3 * FVVDP_LUT_N);
    if (rc != FVVDP_OK) return rc;
    c->h_lut3[tc].assign(h_S_log, h_S_log + n3);
    for (int i = 0; i < FVVDP_LUT_N; ++i) {
        c->h_axes[0][i] = h_Y_log[i];
        c->h_axes[1][i] = h_rho_log[i];
        c->h_axes[2][i] = h_ecc_sqrt[i];
    }
    HIP_TRY(hipMemcpy(c->d_axes, c->h_axes, sizeof(c->h_axes), hipMemcpyHostToDevice));
    c->y_lo = exp2f(h_Y_log[0]);
    c->y_hi = exp2f(h_Y_log[FVVDP_LUT_N - 1]);
    c->sub_valid = false;
    c->rho_lo = exp2f(h_rho_log[0]);
    c->rho_hi = exp2f(h_rho_log[FVVDP_LUT_N - 1]);
    c->ecc_lo = h_ecc_sqrt[0] * h_ecc_sqrt[0];
    c->ecc_hi = h_ecc_sqrt[FVVDP_LUT_N - 1] * h_ecc_sqrt[FVVDP_LUT_N - 1];
    c->lut3_set[tc] = true;
    return FVVDP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// stage 1 launchers
// ------------------------------------------------------------------------------------------------------------
static EotfDev make_eotf(const fvvdp_eotf* e) {
    EotfDev d;
    d.kind = e->kind;
    d.scale = e->Y_peak - e->Y_black;
    d.y_black = e->Y_black;
    d.y_peak = e->Y_peak;
    d.gamma = e->gamma;
    d.l_min = e->L_min;
    d.l_max = e->L_max;
    d.lut = e->d_lut;
    return d;
}

template <int FL, int PX>
static void launch_ring(int dtype, const TemporalArgs& a, hipStream_t st) {
    dim3 grid((a.HW + 256 * PX - 1) / (256 * PX)), block(256);
    if (dtype == FVVDP_U8)
        hipLaunchKernelGGL((temporal_ring_kernel<FL, PX, SRC_U8>), grid, block, 0, st, a);
    else if (dtype == FVVDP_U16)
        hipLaunchKernelGGL((temporal_ring_kernel<FL, PX, SRC_U16>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((temporal_ring_kernel<FL, PX, SRC_F32>), grid, block, 0, st, a);
}

template <int FL, int PX>
static void launch_vec(int dtype, const TemporalArgs& a, hipStream_t st) {
    dim3 grid((a.HW + 64 * PX - 1) / (64 * PX)), block(64);
    if (dtype == FVVDP_U8)
        hipLaunchKernelGGL((temporal_vec_kernel<FL, PX, SRC_U8>), grid, block, 0, st, a);
    else if (dtype == FVVDP_U16)
        hipLaunchKernelGGL((temporal_vec_kernel<FL, PX, SRC_U16>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((temporal_vec_kernel<FL, PX, SRC_F32>), grid, block, 0, st, a);
}

template <int P>
static void launch_generic(int dtype, const GenericArgs& a, hipStream_t st) {
    dim3 grid((a.HW + 255) / 256, a.n_out), block(256);
    if (dtype == FVVDP_U8)
        hipLaunchKernelGGL((temporal_generic_kernel<SRC_U8, P>), grid, block, 0, st, a);
    else if (dtype == FVVDP_U16)
        hipLaunchKernelGGL((temporal_generic_kernel<SRC_U16, P>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((temporal_generic_kernel<SRC_F32, P>), grid, block, 0, st, a);
}

extern "C" int fvvdp_temporal_channels(fvvdp_ctx* c, const void* d_test, const void* d_ref, int dtype, int C,
                                       size_t chan_stride, size_t frame_stride, const fvvdp_eotf* eotf,
                                       const float* h_rgb2y, const int32_t* h_frame_idx, const float* h_taps, int fl,
                                       int n_out, int slot0, int32_t* d_oob_flag, void* stream) {
    if (!c || !d_test || !d_ref || !eotf || !h_frame_idx || !h_taps) return fail(FVVDP_EINVAL, "null argument");
    if (dtype < FVVDP_U8 || dtype > FVVDP_F32) return fail(FVVDP_EINVAL, "Only uint8, uint16 and float32 is currently supported");
    if (C != 1 && C != 3) return fail(FVVDP_EINVAL, "The content must have either 1 or 3 colour channels.");
    if (C == 3 && !h_rgb2y) return fail(FVVDP_EINVAL, "rgb2y weights required for C == 3");
    if (fl < 1 || fl > FVVDP_MAX_TAPS) return fail(FVVDP_EINVAL, "filter length %d out of range", fl);
    if (c->P == 2 && fl != 1) return fail(FVVDP_EINVAL, "still-image context (planes == 2) needs fl == 1");
    if (n_out < 1 || slot0 < 0 || slot0 + n_out > c->max_frames) return fail(FVVDP_EINVAL, "slots [%d,%d) exceed max_frames %d", slot0, slot0 + n_out, c->max_frames);
    if (eotf->kind == FVVDP_EOTF_LUT && (dtype == FVVDP_F32 || !eotf->d_lut)) return fail(FVVDP_EINVAL, "FVVDP_EOTF_LUT needs an integer source and a table");
    if (eotf->kind != FVVDP_EOTF_LUT && dtype != FVVDP_F32) return fail(FVVDP_EINVAL, "integer sources need FVVDP_EOTF_LUT");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int HW = c->W * c->H;
    Timed tm(c, 0, st);
    const bool ring_ok = (c->P == 4) && (fl <= 32);
    if (ring_ok) {
        const int FL = fl <= 8 ? 8 : (fl <= 16 ? 16 : 32);
        const int max_out = T_MAX_IDX - (FL - 1);
        for (int t0 = 0; t0 < n_out; t0 += max_out) {
            const int nn = (n_out - t0) < max_out ? (n_out - t0) : max_out;
            TemporalArgs a;
            memset(&a, 0, sizeof(a));
            a.src[0] = d_test;
            a.src[1] = d_ref;
            a.chan_stride = chan_stride;
            a.frame_stride = frame_stride;
            a.C = C;
            a.HW = HW;
            a.e = make_eotf(eotf);
            if (C == 3) { a.w[0] = h_rgb2y[0]; a.w[1] = h_rgb2y[1]; a.w[2] = h_rgb2y[2]; } else { a.w[0] = 1.0f; }
            a.n_out = nn;
            a.fl = fl;
            a.out = c->level[0] + (size_t)(slot0 + t0) * HW * 4;
            a.oob = d_oob_flag;
            for (int k = 0; k < fl; ++k) { a.taps[0][k] = h_taps[k]; a.taps[1][k] = h_taps[fl + k]; }
            // virtual time of h_frame_idx: entry (fl-1+t) is the newest frame of output t; pad older history
            const int pad = FL - fl;
            for (int u = 0; u < FL - 1 + nn; ++u) {
                const int src = t0 + u - pad;         // index into h_frame_idx
                a.idx[u] = h_frame_idx[src < 0 ? 0 : src];
            }
            // vector path needs the lane's PX consecutive samples to be naturally aligned
            const int PXv = FL == 32 ? 2 : 4;
            const int es = dtype == FVVDP_U8 ? 1 : (dtype == FVVDP_U16 ? 2 : 4);
            const bool vec_ok = !getenv("FVVDP_TEMPORAL_SCALAR") && (HW % PXv == 0) && (HW >= PXv) &&
                                (chan_stride % PXv == 0) && (frame_stride % PXv == 0) &&
                                (reinterpret_cast<uintptr_t>(d_test) % (size_t)(es * PXv) == 0) &&
                                (reinterpret_cast<uintptr_t>(d_ref) % (size_t)(es * PXv) == 0);
            if (vec_ok) {
                if (FL == 8) launch_vec<8, 4>(dtype, a, st);
                else if (FL == 16) launch_vec<16, 4>(dtype, a, st);
                else launch_vec<32, 2>(dtype, a, st);
            } else {
                if (FL == 8) launch_ring<8, 4>(dtype, a, st);
                else if (FL == 16) launch_ring<16, 4>(dtype, a, st);
                else launch_ring<32, 2>(dtype, a, st);
            }
        }
    } else {
        // rare path: tables go through device buffers, uploaded synchronously
        if (fl - 1 + n_out > c->max_frames + FVVDP_MAX_TAPS) return fail(FVVDP_EINVAL, "too many frames for one call");
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(c->d_taps, h_taps, sizeof(float) * 2 * fl, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_idx, h_frame_idx, sizeof(int) * (fl - 1 + n_out), hipMemcpyHostToDevice));
        GenericArgs a;
        memset(&a, 0, sizeof(a));
        a.src[0] = d_test;
        a.src[1] = d_ref;
        a.chan_stride = chan_stride;
        a.frame_stride = frame_stride;
        a.C = C;
        a.HW = HW;
        a.e = make_eotf(eotf);
        if (C == 3) { a.w[0] = h_rgb2y[0]; a.w[1] = h_rgb2y[1]; a.w[2] = h_rgb2y[2]; } else { a.w[0] = 1.0f; }
        a.n_out = n_out;
        a.fl = fl;
        a.out = c->level[0] + (size_t)slot0 * HW * c->P;
        a.oob = d_oob_flag;
        a.taps = c->d_taps;
        a.idx = c->d_idx;
        if (c->P == 2) launch_generic<2>(dtype, a, st);
        else launch_generic<4>(dtype, a, st);
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

template <int FL, int PX>
static void launch_yuv(int bytes, const YuvArgs& a, hipStream_t st) {
    const int HW = a.W * a.H;
    dim3 grid((HW + 256 * PX - 1) / (256 * PX)), block(256);
    if (bytes == 1) hipLaunchKernelGGL((temporal_yuv_kernel<FL, PX, unsigned char>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((temporal_yuv_kernel<FL, PX, unsigned short>), grid, block, 0, st, a);
}

extern "C" int fvvdp_temporal_channels_yuv(fvvdp_ctx* c, const void* d_test, const void* d_ref, const fvvdp_yuv_format* fmt,
                                           size_t frame_stride, const fvvdp_eotf* eotf, const float* h_rgb2y,
                                           const int32_t* h_frame_idx, const float* h_taps, int fl, int n_out, int slot0,
                                           int32_t* d_oob_flag, void* stream) {
    if (!c || !d_test || !d_ref || !fmt || !eotf || !h_rgb2y || !h_frame_idx || !h_taps) return fail(FVVDP_EINVAL, "null argument");
    if (c->P != 4) return fail(FVVDP_EINVAL, "YUV ingest is for video contexts (planes == 4)");
    if (fmt->bit_depth < 8 || fmt->bit_depth > 16) return fail(FVVDP_EINVAL, "bit depth %d not supported", fmt->bit_depth);
    if (fmt->chroma_420 && ((c->W | c->H) & 1)) return fail(FVVDP_EINVAL, "4:2:0 needs even frame dimensions");
    if (fl < 1 || fl > 32) return fail(FVVDP_EINVAL, "filter length %d out of range for the YUV path (1..32)", fl);
    if (n_out < 1 || slot0 < 0 || slot0 + n_out > c->max_frames) return fail(FVVDP_EINVAL, "slots out of range");
    if (eotf->kind == FVVDP_EOTF_LUT) return fail(FVVDP_EINVAL, "YUV sources need a closed-form display model (RGB is fractional after the matrix)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    Timed tm(c, 0, st);
    const int FL = fl <= 8 ? 8 : (fl <= 16 ? 16 : 32);
    const int max_out = T_MAX_IDX - (FL - 1);
    for (int t0 = 0; t0 < n_out; t0 += max_out) {
        const int nn = (n_out - t0) < max_out ? (n_out - t0) : max_out;
        YuvArgs a;
        memset(&a, 0, sizeof(a));
        a.src[0] = d_test;
        a.src[1] = d_ref;
        a.frame_stride = frame_stride;
        a.W = c->W;
        a.H = c->H;
        a.chroma420 = fmt->chroma_420 ? 1 : 0;
        a.uvw = fmt->chroma_420 ? c->W / 2 : c->W;
        a.uvh = fmt->chroma_420 ? c->H / 2 : c->H;
        const float scale = (float)(1 << (fmt->bit_depth - 8));
        a.wy = 1.0f / (scale * 219.0f);
        a.wc = 1.0f / (scale * 224.0f);
        for (int i = 0; i < 9; ++i) a.m[i] = fmt->ycbcr2rgb[i];
        a.e = make_eotf(eotf);
        a.w[0] = h_rgb2y[0]; a.w[1] = h_rgb2y[1]; a.w[2] = h_rgb2y[2];
        a.n_out = nn;
        a.fl = fl;
        a.out = c->level[0] + (size_t)(slot0 + t0) * c->W * c->H * 4;
        a.oob = d_oob_flag;
        for (int k = 0; k < fl; ++k) { a.taps[0][k] = h_taps[k]; a.taps[1][k] = h_taps[fl + k]; }
        const int pad = FL - fl;
        for (int u = 0; u < FL - 1 + nn; ++u) {
            const int src = t0 + u - pad;
            a.idx[u] = h_frame_idx[src < 0 ? 0 : src];
        }
        const int bytes = fmt->bit_depth > 8 ? 2 : 1;
        if (FL == 8) launch_yuv<8, 2>(bytes, a, st);
        else if (FL == 16) launch_yuv<16, 2>(bytes, a, st);
        else launch_yuv<32, 1>(bytes, a, st);
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_load_channels_planar(fvvdp_ctx* c, const float* d_R, int n, int slot0, void* stream) {
    if (!c || !d_R) return fail(FVVDP_EINVAL, "null argument");
    if (n < 1 || slot0 < 0 || slot0 + n > c->max_frames) return fail(FVVDP_EINVAL, "slots out of range");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int HW = c->W * c->H;
    dim3 grid((HW + 255) / 256, n), block(256);
    float* out = c->level[0] + (size_t)slot0 * HW * c->P;
    if (c->P == 4) hipLaunchKernelGGL((interleave_kernel<4>), grid, block, 0, st, d_R, out, HW, 1);
    else hipLaunchKernelGGL((interleave_kernel<2>), grid, block, 0, st, d_R, out, HW, 1);
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_export_level(fvvdp_ctx* c, int level, int n, float* d_out, void* stream) {
    if (!c || !d_out) return fail(FVVDP_EINVAL, "null argument");
    if (level < 0 || level > c->n_bands || n < 1 || n > c->max_frames) return fail(FVVDP_EINVAL, "bad level/n");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int HW = c->lw[level] * c->lh[level];
    dim3 grid((HW + 255) / 256, n), block(256);
    if (c->P == 4) hipLaunchKernelGGL((interleave_kernel<4>), grid, block, 0, st, c->level[level], d_out, HW, 0);
    else hipLaunchKernelGGL((interleave_kernel<2>), grid, block, 0, st, c->level[level], d_out, HW, 0);
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// stage 2 launcher
// ------------------------------------------------------------------------------------------------------------
// Foveated mode: slice of the 32^3 LUT that band b can reach.  rho = rho_band*res_mag with res_mag in
// [1, res_mag(corner of the screen)], so only a few rho knots are live per band; the slice is stored as
// [ecc][Y][rho] of float4 {S0[i], S1[i], S0[i+1], S1[i+1]} so that one aligned 16-byte load returns the two rho
// corners of both temporal channels and neighbouring pixels (similar ecc, Y) share cache lines.
static int build_sublut(fvvdp_ctx* c, const fvvdp_geom* g) {
    fvvdp_geom key;
    memset(&key, 0, sizeof(key));
    if (g) key = *g;
    if (c->sub_valid && memcmp(&c->sub_geom, &key, sizeof(key)) == 0) return FVVDP_OK;
    double mag_max = 1.0, mag_min = 1.0;
    if (g) {
        const double delta = (1.0 / (double)g->ppd_centre) / 2.0 * M_PI / 180.0;
        const double ax = atan(0.5 * g->display_size_m[0] / g->distance_m) * 180.0 / M_PI;
        const double ay = atan(0.5 * g->display_size_m[1] / g->distance_m) * 180.0 / M_PI;
        double va = sqrt(ax * ax + ay * ay);
        if (va > 89.9) va = 89.9;
        va *= M_PI / 180.0;
        mag_max = cos(delta) / (cos(va) * cos(va + delta)) * 1.01;   // 1 % margin over the corner pixel
    }
    const float* xr = c->h_axes[1];
    for (int b = 0; b < c->n_bands; ++b) {
        if (!g) {                // user geometry: the caller states the range of its magnification map
            mag_max = c->map_rm_max[b] > 0 ? (double)c->map_rm_max[b] * 1.01 : 1e9;
            mag_min = c->map_rm_min[b] > 0 ? (double)c->map_rm_min[b] * 0.99 : 0.0;
        }
        double r_lo = c->rho_band[b] * mag_min, r_hi = c->rho_band[b] * mag_max;
        const double lo_c = exp2((double)xr[0]), hi_c = exp2((double)xr[FVVDP_LUT_N - 1]);
        r_lo = fmin(fmax(r_lo, lo_c), hi_c);
        r_hi = fmin(fmax(r_hi, lo_c), hi_c);
        int i_lo = 0, i_hi = FVVDP_LUT_N - 1;
        while (i_lo + 1 < FVVDP_LUT_N - 1 && xr[i_lo + 1] < log2(r_lo) - 1e-3) ++i_lo;   // last knot <= log2(r_lo)
        while (i_hi - 1 > i_lo && xr[i_hi - 1] > log2(r_hi) + 1e-3) --i_hi;              // first knot >= log2(r_hi)
        const int rw = i_hi - i_lo;                                                       // intervals covered
        if (c->sublut[b] && c->sub_rw[b] != rw) {
            (void)hipFree(c->sublut[b]);
            c->sublut[b] = nullptr;
        }
        const size_t n = (size_t)FVVDP_LUT_N * FVVDP_LUT_N * rw;
        if (!c->sublut[b]) {
            int rc = dev_alloc(c, &c->sublut[b], n);
            if (rc != FVVDP_OK) return rc;
        }
        std::vector<float4> h(n);
        const float* L0 = c->h_lut3[0].data();
        const float* L1 = c->h_lut3[1].empty() ? L0 : c->h_lut3[1].data();
        for (int k = 0; k < FVVDP_LUT_N; ++k)
            for (int j = 0; j < FVVDP_LUT_N; ++j)
                for (int i = 0; i < rw; ++i) {
                    const size_t s0 = ((size_t)j * FVVDP_LUT_N + (i_lo + i)) * FVVDP_LUT_N + k;       // [Y][rho][ecc]
                    const size_t s1 = ((size_t)j * FVVDP_LUT_N + (i_lo + i + 1)) * FVVDP_LUT_N + k;
                    h[((size_t)k * FVVDP_LUT_N + j) * rw + i] = make_float4(L0[s0], L1[s0], L0[s1], L1[s1]);
                }
        HIP_TRY(hipMemcpy(c->sublut[b], h.data(), n * sizeof(float4), hipMemcpyHostToDevice));
        c->sub_rw[b] = rw;
        c->sub_ilo[b] = i_lo;
    }
    c->sub_geom = key;
    c->sub_valid = true;
    return FVVDP_OK;
}

template <int P>
static void launch_band(const BandArgs& a, int nblocks, bool dbg, bool fov, hipStream_t st) {
    dim3 grid(nblocks), block(64);
    if (fov) {
        const dim3 gridf((nblocks + FOV_WPB - 1) / FOV_WPB), blockf(64 * FOV_WPB);
        const size_t lds = a.lut_lds ? (size_t)FVVDP_LUT_N * FVVDP_LUT_N * a.rw * sizeof(float4) : 0;
        if (dbg) hipLaunchKernelGGL((band_kernel<P, true, true>), gridf, blockf, lds, st, a);
        else hipLaunchKernelGGL((band_kernel<P, false, true>), gridf, blockf, lds, st, a);
    } else {
        if (dbg) hipLaunchKernelGGL((band_kernel<P, true, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((band_kernel<P, false, false>), grid, block, 0, st, a);
    }
}

extern "C" int fvvdp_bands_forward(fvvdp_ctx* c, int n, float* d_Q, int q_stride, int q_col0, const float* h_fixation,
                                   const fvvdp_geom* geom, const fvvdp_band_maps* maps, void* stream) {
    if (!c || !d_Q) return fail(FVVDP_EINVAL, "null argument");
    if (n < 1 || n > c->max_frames) return fail(FVVDP_EINVAL, "n=%d exceeds max_frames=%d", n, c->max_frames);
    if (q_col0 < 0 || q_col0 + n > q_stride) return fail(FVVDP_EINVAL, "Q columns out of range");
    const bool fov = h_fixation != nullptr;
    if (fov && !geom && !c->maps_set) return fail(FVVDP_EINVAL, "foveated mode needs the display geometry (or fvvdp_ctx_set_view_maps)");
    if (fov && !(c->lut3_set[0] && (c->P == 2 || c->lut3_set[1]))) return fail(FVVDP_ESTATE, "fvvdp_ctx_set_csf_3d not called");
    if (!fov && !c->csf_set) return fail(FVVDP_ESTATE, "fvvdp_ctx_set_csf_1d not called");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (fov) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(c->d_fix, h_fixation, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
        int rc = build_sublut(c, geom);
        if (rc != FVVDP_OK) return rc;
    }
    FinalizeArgs fa;
    memset(&fa, 0, sizeof(fa));
    for (int b = 0; b < c->n_bands; ++b) {
        BandArgs a;
        memset(&a, 0, sizeof(a));
        a.Gf = c->level[b];
        a.Gc = c->level[b + 1];
        a.w = c->lw[b];
        a.h = c->lh[b];
        a.wc = c->lw[b + 1];
        a.hc = c->lh[b + 1];
        a.n_strips = band_strips(a.wc);
        chunking(a.hc, a.n_strips, n, c->wave_capacity, a.n_chunks, a.cr);
        a.band_mul = (b == 0) ? 1.0f : 2.0f;                 // lpyr.get_band, fvvdp_lpyr_dec.py:57-63
        a.csf = c->csf + (size_t)b * FVVDP_LUT_N;
        a.csf_y = c->csf_y;
        a.y_first = c->y_first;
        a.y_inv_step = c->y_inv_step;
        a.y_lo = c->y_lo;
        a.y_hi = c->y_hi;
        a.ly_lo = log2f(c->y_lo);
        a.ly_hi = log2f(c->y_hi);
        a.lg_gain = log2f(c->prm.sens_gain);
        a.lg_k = log2f(c->prm.mask_k);
        a.p = c->prm.mask_p;
        a.q0 = c->prm.mask_q[0];
        a.q1 = c->prm.mask_q[1];
        a.beta = c->prm.beta;
        a.lbkg_min = c->prm.lbkg_min;
        a.cmax = c->prm.contrast_max;
        a.lg_dmax = log2f(c->prm.d_max);
        a.partial = c->partial + c->partial_off[b];
        bool dbg = false;
        if (maps) {
            a.dD = maps[b].d_D;
            a.dC = maps[b].d_contrast;
            a.dL = maps[b].d_lbkg;
            a.dS = maps[b].d_S;
            dbg = a.dD || a.dC || a.dL || a.dS;
        }
        if (fov) {
            a.sublut = c->sublut[b];
            a.axes = c->d_axes;
            a.rw = c->sub_rw[b];
            a.i_lo = c->sub_ilo[b];
            a.fix = c->d_fix;
            if (geom) {
                a.size_m0 = geom->display_size_m[0];
                a.size_m1 = geom->display_size_m[1];
                a.dist_m = geom->distance_m;
                const double delta = (1.0 / (double)geom->ppd_centre) / 2.0 * M_PI / 180.0;
                a.delta_rad = (float)delta;
                a.cos_delta = (float)cos(delta);
            } else {
                a.size_m0 = a.size_m1 = a.dist_m = 1.0f;
                a.mvx = c->map_vx[b];
                a.mvy = c->map_vy[b];
                a.mrm = c->map_rm[b];
            }
            a.rho_band = (float)c->rho_band[b];
            a.rho_lo = c->rho_lo;
            a.rho_hi = c->rho_hi;
            a.ecc_lo = c->ecc_lo;
            a.ecc_hi = c->ecc_hi;
            for (int ax = 0; ax < 3; ++ax) {
                a.first[ax] = c->h_axes[ax][0];
                a.inv_step[ax] = (float)(FVVDP_LUT_N - 1) / (c->h_axes[ax][FVVDP_LUT_N - 1] - c->h_axes[ax][0]);
            }
            a.frame_w = c->W;
            a.frame_h = c->H;
        }
        const int nblk = a.n_strips * a.n_chunks;
        if (nblk > c->max_blk[b]) return fail(FVVDP_ESTATE, "internal: partial buffer too small");
        a.n_items = nblk * n;
        a.lut_lds = (fov && (size_t)FVVDP_LUT_N * FVVDP_LUT_N * c->sub_rw[b] * sizeof(float4) <= 48 * 1024) ? 1 : 0;
        {
            Timed tm(c, 1 + b, st);
            if (c->P == 4) launch_band<4>(a, nblk * n, dbg, fov, st);
            else launch_band<2>(a, nblk * n, dbg, fov, st);
        }
        fa.nblk[b] = nblk;
        fa.off[b] = c->partial_off[b];
        fa.npx[b] = (float)a.w * (float)a.h;
    }
    fa.partial = c->partial;
    fa.Q = d_Q;
    fa.n_bands = c->n_bands;
    fa.n = n;
    fa.q_stride = q_stride;
    fa.q_col0 = q_col0;
    fa.tc = c->P / 2;
    fa.inv_beta = 1.0f / c->prm.beta;
    {
        Timed tm(c, 1 + c->n_bands, st);
        const int total = c->n_bands * 2 * n;
        hipLaunchKernelGGL(finalize_kernel, dim3(total), dim3(64), 0, st, fa);
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_ctx_set_view_maps(fvvdp_ctx* c, int band, const float* d_view_x, const float* d_view_y,
                                       const float* d_res_mag, float res_mag_min, float res_mag_max) {
    if (!c) return fail(FVVDP_EINVAL, "null context");
    if (band < 0 || band >= c->n_bands) return fail(FVVDP_EINVAL, "band %d out of range", band);
    if (!d_view_x || !d_view_y || !d_res_mag) {      // clear
        for (int b = 0; b < FVVDP_MAX_BANDS; ++b) c->map_vx[b] = c->map_vy[b] = c->map_rm[b] = nullptr;
        c->maps_set = false;
        c->sub_valid = false;
        return FVVDP_OK;
    }
    c->map_vx[band] = d_view_x;
    c->map_vy[band] = d_view_y;
    c->map_rm[band] = d_res_mag;
    c->map_rm_max[band] = res_mag_max;
    c->map_rm_min[band] = res_mag_min;
    c->maps_set = true;
    for (int b = 0; b < c->n_bands; ++b)
        if (!c->map_vx[b]) c->maps_set = false;      // complete only when every band has its maps
    c->sub_valid = false;
    return FVVDP_OK;
}

extern "C" int fvvdp_heatmap_reconstruct(fvvdp_ctx* c, int n, const float* const* h_dD, float w_transient, float beta_jod,
                                         float jod_a_abs, float* d_out, void* stream) {
    if (!c || !h_dD || !d_out) return fail(FVVDP_EINVAL, "null argument");
    if (n < 1 || n > c->max_frames) return fail(FVVDP_EINVAL, "n=%d exceeds max_frames=%d", n, c->max_frames);
    for (int b = 0; b < c->n_bands; ++b)
        if (!h_dD[b]) return fail(FVVDP_EINVAL, "missing D map of band %d", b);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    for (int b = 1; b < c->n_bands; ++b)
        if (!c->heat[b]) {
            int rc = dev_alloc(c, &c->heat[b], (size_t)c->max_frames * c->lw[b] * c->lh[b]);
            if (rc != FVVDP_OK) return rc;
        }
    for (int b = c->n_bands - 1; b >= 0; --b) {
        HeatArgs a;
        memset(&a, 0, sizeof(a));
        a.D = h_dD[b];
        a.coarse = (b == c->n_bands - 1) ? nullptr : c->heat[b + 1];   // the base band of a zero image is zero
        a.out = (b == 0) ? d_out : c->heat[b];
        a.w = c->lw[b];
        a.h = c->lh[b];
        a.wc = c->lw[b + 1];
        a.hc = c->lh[b + 1];
        a.tc = c->P / 2;
        a.w_trans = w_transient;
        a.inv_m = (b == 0) ? 1.0f : 0.5f;                              // set_band divides by the band multiplier
        a.beta_jod = beta_jod;
        a.scale = jod_a_abs;
        a.final_level = (b == 0);
        dim3 grid((a.w + 255) / 256, a.h, n), block(256);
        hipLaunchKernelGGL(heat_level_kernel, grid, block, 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return FVVDP_OK;
}

extern "C" int fvvdp_ctx_timing_enable(fvvdp_ctx* c, int on) {
    if (!c) return fail(FVVDP_EINVAL, "null context");
    c->timing = on != 0;
    return FVVDP_OK;
}

extern "C" int fvvdp_ctx_timing_read(fvvdp_ctx* c, float* h_ms, int32_t* h_count, int capacity, int reset) {
    if (!c || !h_ms || !h_count) return fail(FVVDP_EINVAL, "null argument");
    const int nk = c->n_bands + 2;
    if (capacity < nk) return fail(FVVDP_EINVAL, "capacity %d < %d kernels", capacity, nk);
    for (int id = 0; id < nk; ++id) {
        for (auto& pr : c->ev[id]) {
            HIP_TRY(hipEventSynchronize(pr.second));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
            c->t_ms[id] += ms;
            c->t_cnt[id] += 1;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        c->ev[id].clear();
        h_ms[id] = c->t_ms[id];
        h_count[id] = c->t_cnt[id];
        if (reset) {
            c->t_ms[id] = 0.0f;
            c->t_cnt[id] = 0;
        }
    }
    return FVVDP_OK;
}
