// Stage 1 kernels: unpack + display photometry + luminance + temporal filter -> pyramid level 0.
// Included by fvvdp_hip.hip (one translation unit).
#pragma once
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
template <int KIND>
__device__ __forceinline__ float eotf_one(float V, const EotfDev& e, bool& bad) {
    if constexpr (KIND == FVVDP_EOTF_SRGB) {
        bad = bad || (V > 1.0f) || (V < 0.0f);
        V = fminf(fmaxf(V, 0.0f), 1.0f);
        // constant divisions as reciprocal multiplies (<= 1 ulp, below the error of the fast log2/exp2 pair)
        const float hi = fast_exp2(2.4f * fast_log2((V + 0.055f) * (1.0f / 1.055f)));
        const float lin = V > 0.04045f ? hi : V * (1.0f / 12.92f);
        return __fadd_rn(__fmul_rn(e.scale, lin), e.y_black);
    } else if constexpr (KIND == FVVDP_EOTF_GAMMA) {
        bad = bad || (V > 1.0f) || (V < 0.0f);
        V = fminf(fmaxf(V, 0.0f), 1.0f);
        const float lin = V > 0.0f ? fast_exp2(e.gamma * fast_log2(V)) : 0.0f;
        return __fadd_rn(__fmul_rn(e.scale, lin), e.y_black);
    } else if constexpr (KIND == FVVDP_EOTF_PQ) {
        bad = bad || (V > 1.0f) || (V < 0.0f);
        V = fminf(fmaxf(V, 0.0f), 1.0f);
        const float m = 78.843750000000000f, n = 0.15930175781250000f;
        const float c1 = 0.83593750000000000f, c2 = 18.851562500000000f, c3 = 18.687500000000000f;
        const float im_t = V > 0.0f ? fast_exp2(fast_log2(V) * (1.0f / m)) : 0.0f;
        const float r = fmaxf(im_t - c1, 0.0f) * __builtin_amdgcn_rcpf(c2 - c3 * im_t);
        const float L = r > 0.0f ? 10000.0f * fast_exp2(fast_log2(r) * (1.0f / n)) : 0.0f;
        return fminf(fmaxf(L, 0.005f), e.y_peak) + e.y_black;
    } else if constexpr (KIND == FVVDP_EOTF_LINEAR) {
        return fminf(fmaxf(V, 0.005f), e.y_peak) + e.y_black;
    } else if constexpr (KIND == FVVDP_EOTF_ABSOLUTE) {
        return fminf(fmaxf(V, e.l_min), e.l_max);
    } else {
        return V;
    }
}

__device__ __forceinline__ float eotf_f32(float V, const EotfDev& e, bool& bad) {
    switch (e.kind) {
        case FVVDP_EOTF_SRGB: return eotf_one<FVVDP_EOTF_SRGB>(V, e, bad);
        case FVVDP_EOTF_GAMMA: return eotf_one<FVVDP_EOTF_GAMMA>(V, e, bad);
        case FVVDP_EOTF_PQ: return eotf_one<FVVDP_EOTF_PQ>(V, e, bad);
        case FVVDP_EOTF_LINEAR: return eotf_one<FVVDP_EOTF_LINEAR>(V, e, bad);
        case FVVDP_EOTF_ABSOLUTE: return eotf_one<FVVDP_EOTF_ABSOLUTE>(V, e, bad);
        default: return V;
    }
}

// N samples at once with ONE (wave-uniform) branch on the display model: in the register-ring kernels the per-sample
// switch of eotf_f32 is replicated FL x PX x 3 x 2 times, which made their code several times larger than the
// instruction cache although only one case ever runs.
template <int N>
__device__ __forceinline__ void eotf_apply(float (&V)[N], const EotfDev& e, bool& bad) {
#define FVVDP_EOTF_CASE(K)                                             \
    case K: {                                                          \
        _Pragma("unroll") for (int i = 0; i < N; ++i) V[i] = eotf_one<K>(V[i], e, bad); \
        break;                                                         \
    }
    switch (e.kind) {
        FVVDP_EOTF_CASE(FVVDP_EOTF_SRGB)
        FVVDP_EOTF_CASE(FVVDP_EOTF_GAMMA)
        FVVDP_EOTF_CASE(FVVDP_EOTF_PQ)
        FVVDP_EOTF_CASE(FVVDP_EOTF_LINEAR)
        FVVDP_EOTF_CASE(FVVDP_EOTF_ABSOLUTE)
        default: break;
    }
#undef FVVDP_EOTF_CASE
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
    if constexpr (SRC == SRC_F32) {
        if (C == 3) {
            float t[3 * PX];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < PX; ++i) t[c * PX + i] = f.ch[c].value(i);
            eotf_apply<3 * PX>(t, e, bad);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < PX; ++i) v[c][i] = __fmul_rn(t[c * PX + i], w[c]);
        } else {
            float t[PX];
#pragma unroll
            for (int i = 0; i < PX; ++i) t[i] = f.ch[0].value(i);
            eotf_apply<PX>(t, e, bad);
#pragma unroll
            for (int i = 0; i < PX; ++i) v[0][i] = __fmul_rn(t[i], w[0]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c > 0 && C != 3) break;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                if constexpr (SRC == SRC_U8) v[c][i] = lutw[c * 256 + f.ch[c].code(i)];
                else v[c][i] = __fmul_rn(lut16[f.ch[c].code(i)], w[c]);
            }
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

// ---- vector variant of the YUV ingest (the fast path for W % 4 == 0) -----------------------------------------------
// Single-wave workgroups, a lane owns 4 CONSECUTIVE pixels of one image row (x0 = 4j).  Everything that depends only
// on the pixel position (plane offsets, the bilinear weights of the 4:2:0 chroma upsampling) is computed once per
// lane; per frame a lane issues one Y load (4 samples) and, per chroma plane and source row, one aligned pair load
// (columns 2j, 2j+1) plus the two neighbour columns (2j-1, 2j+2, clamped) -- 13 loads for 4 pixels instead of 36.
// The arithmetic (order of the products and sums) is the one of yuv_lum above, so both kernels agree bit for bit;
// raw samples of the next frame are prefetched while the current one is converted, and the finished float4 pixels
// go through the same LDS transpose as temporal_vec_kernel.
template <typename T, bool C420>
struct YuvRaw {
    static constexpr int YW = (int)sizeof(T);          // dwords holding 4 samples: 1 (8 bit) or 2 (10..16 bit)
    unsigned int y[YW];
    // 4:2:0: [plane][row] pair = columns 2j,2j+1 packed; lft / rgt = neighbour columns.  4:4:4: cp[plane][0..YW-1].
    unsigned int cp[2][2], lft[2][2], rgt[2][2];
    __device__ __forceinline__ float ysample(int i) const {
        if constexpr (sizeof(T) == 1) return (float)((y[0] >> (8 * i)) & 0xFFu);
        else return (float)((y[i / 2] >> (16 * (i % 2))) & 0xFFFFu);
    }
    __device__ __forceinline__ float c444(int pl, int i) const {
        if constexpr (sizeof(T) == 1) return (float)((cp[pl][0] >> (8 * i)) & 0xFFu);
        else return (float)((cp[pl][i / 2] >> (16 * (i % 2))) & 0xFFFFu);
    }
    __device__ __forceinline__ float pair(int pl, int r, int k) const {
        if constexpr (sizeof(T) == 1) return (float)((cp[pl][r] >> (8 * k)) & 0xFFu);
        else return (float)((cp[pl][r] >> (16 * k)) & 0xFFFFu);
    }
};

struct YuvGeom {           // per-lane constants
    int oy;                // element offset of the 4 luma samples inside a frame
    int opair[2];          // offsets (from the start of a chroma plane) of the pair in the two source rows
    int oleft[2], oright[2];
    float fy, gy;          // vertical weights
    float fx0, gx0;        // horizontal weights of pixel 0 (0/1 at the left image edge, else .75/.25)
};

template <typename T, bool C420>
__device__ __forceinline__ YuvRaw<T, C420> yuv_fetch(const T* __restrict__ f, const YuvGeom& g, int HW, int uvplane) {
    YuvRaw<T, C420> r;
    if constexpr (sizeof(T) == 1) {
        r.y[0] = *reinterpret_cast<const unsigned int*>(f + g.oy);
    } else {
        const uint2 t = *reinterpret_cast<const uint2*>(f + g.oy);
        r.y[0] = t.x; r.y[1] = t.y;
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const T* P = f + HW + pl * uvplane;
        if constexpr (C420) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                if constexpr (sizeof(T) == 1) r.cp[pl][rr] = *reinterpret_cast<const unsigned short*>(P + g.opair[rr]);
                else r.cp[pl][rr] = *reinterpret_cast<const unsigned int*>(P + g.opair[rr]);
                r.lft[pl][rr] = P[g.oleft[rr]];
                r.rgt[pl][rr] = P[g.oright[rr]];
            }
        } else {
            if constexpr (sizeof(T) == 1) {
                r.cp[pl][0] = *reinterpret_cast<const unsigned int*>(P + g.oy);
            } else {
                const uint2 t = *reinterpret_cast<const uint2*>(P + g.oy);
                r.cp[pl][0] = t.x; r.cp[pl][1] = t.y;
            }
        }
    }
    return r;
}

template <typename T, bool C420>
__device__ __forceinline__ void yuv_vec_rgb(const YuvRaw<T, C420>& r, const YuvArgs& a, const YuvGeom& g, float (&rgb)[12]) {
    auto cf = [&](float code) { return fminf(fmaxf(a.wc * code - (128.0f / 224.0f), -0.5f), 0.5f); };
    float uv[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        if constexpr (C420) {
            float hrow[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const float cl = cf((float)r.lft[pl][rr]), c0 = cf(r.pair(pl, rr, 0)), c1 = cf(r.pair(pl, rr, 1)), cr = cf((float)r.rgt[pl][rr]);
                hrow[rr][0] = g.gx0 * cl + g.fx0 * c0;
                hrow[rr][1] = 0.75f * c0 + 0.25f * c1;
                hrow[rr][2] = 0.25f * c0 + 0.75f * c1;
                hrow[rr][3] = 0.75f * c1 + 0.25f * cr;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) uv[pl][i] = g.gy * hrow[0][i] + g.fy * hrow[1][i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) uv[pl][i] = cf(r.c444(pl, i));
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float Yf = fminf(fmaxf(a.wy * r.ysample(i) - (16.0f / 219.0f), 0.0f), 1.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = a.m[3 * c] * Yf + a.m[3 * c + 1] * uv[0][i] + a.m[3 * c + 2] * uv[1][i];
            rgb[3 * i + c] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
    }
}

// display model + luminance of the 4 pixels of both streams (rgb[s][3*i+c])
__device__ __forceinline__ void yuv_vec_lum2(float (&rgb)[24], const YuvArgs& a, float (&L0)[4], float (&L1)[4], bool& bad) {
    eotf_apply<24>(rgb, a.e, bad);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        L0[i] = __fadd_rn(__fadd_rn(__fmul_rn(rgb[3 * i], a.w[0]), __fmul_rn(rgb[3 * i + 1], a.w[1])), __fmul_rn(rgb[3 * i + 2], a.w[2]));
        L1[i] = __fadd_rn(__fadd_rn(__fmul_rn(rgb[12 + 3 * i], a.w[0]), __fmul_rn(rgb[12 + 3 * i + 1], a.w[1])),
                          __fmul_rn(rgb[12 + 3 * i + 2], a.w[2]));
    }
}

template <int FL, typename T, bool C420>
__global__ __launch_bounds__(64) void temporal_yuv_vec_kernel(const YuvArgs a) {
    constexpr int PX = 4;
    __shared__ float4 s_t[64 * (PX + 1)];
    const int lane = threadIdx.x;
    const int HW = a.W * a.H;
    const int uvplane = a.uvw * a.uvh;
    const int p0 = blockIdx.x * (64 * PX);
    const int pl = min(p0 + lane * PX, HW - PX);
    YuvGeom g;
    {
        const int y = pl / a.W, x = pl - y * a.W;
        g.oy = pl;
        if constexpr (C420) {
            // torch bilinear, align_corners=False: source = (dst + 0.5)/2 - 0.5 clamped at 0 (video_source_file.py:262-266)
            const float sy = fmaxf(((float)y + 0.5f) * 0.5f - 0.5f, 0.0f);
            const int y0 = (int)sy, y1 = min(y0 + 1, a.uvh - 1);
            g.fy = sy - (float)y0;
            g.gy = 1.0f - g.fy;
            const int j2 = x >> 1;                       // column of the aligned pair
            const int cl = max(j2 - 1, 0), cr = min(j2 + 2, a.uvw - 1);
            g.opair[0] = y0 * a.uvw + j2;  g.opair[1] = y1 * a.uvw + j2;
            g.oleft[0] = y0 * a.uvw + cl;  g.oleft[1] = y1 * a.uvw + cl;
            g.oright[0] = y0 * a.uvw + cr; g.oright[1] = y1 * a.uvw + cr;
            const float sx = fmaxf(((float)x + 0.5f) * 0.5f - 0.5f, 0.0f);
            g.fx0 = sx - (float)(int)sx;                 // 0 at x == 0 (then the "left" column is column 0 itself), else .75
            g.gx0 = 1.0f - g.fx0;
        } else {
            g.fy = g.gy = g.fx0 = g.gx0 = 0.0f;
            g.opair[0] = g.opair[1] = g.oleft[0] = g.oleft[1] = g.oright[0] = g.oright[1] = 0;
        }
    }
    bool bad = false;
    float ring[2][FL][PX];
#pragma unroll
    for (int u = 0; u < FL; ++u)
#pragma unroll
        for (int i = 0; i < PX; ++i) ring[0][u][i] = ring[1][u][i] = 0.0f;
    const int total = FL - 1 + a.n_out;
    YuvRaw<T, C420> nx[2];
    {
        const size_t off = (size_t)a.idx[0] * a.frame_stride;
        nx[0] = yuv_fetch<T, C420>(reinterpret_cast<const T*>(a.src[0]) + off, g, HW, uvplane);
        nx[1] = yuv_fetch<T, C420>(reinterpret_cast<const T*>(a.src[1]) + off, g, HW, uvplane);
    }
    for (int v0 = 0; v0 < total; v0 += FL) {
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int v = v0 + u;
            if (v < total) {
                const YuvRaw<T, C420> cur0 = nx[0], cur1 = nx[1];
                if (v + 1 < total) {
                    const size_t off = (size_t)a.idx[v + 1] * a.frame_stride;
                    nx[0] = yuv_fetch<T, C420>(reinterpret_cast<const T*>(a.src[0]) + off, g, HW, uvplane);
                    nx[1] = yuv_fetch<T, C420>(reinterpret_cast<const T*>(a.src[1]) + off, g, HW, uvplane);
                }
                float rgb[24];
                {
                    float t0[12], t1[12];
                    yuv_vec_rgb<T, C420>(cur0, a, g, t0);
                    yuv_vec_rgb<T, C420>(cur1, a, g, t1);
#pragma unroll
                    for (int i = 0; i < 12; ++i) { rgb[i] = t0[i]; rgb[12 + i] = t1[i]; }
                }
                yuv_vec_lum2(rgb, a, ring[0][u], ring[1][u], bad);
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
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < PX; ++i)
                        s_t[lane * (PX + 1) + i] = make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
                    __syncthreads();
                    float4* o = reinterpret_cast<float4*>(a.out) + (size_t)(v - (FL - 1)) * HW + p0;
#pragma unroll
                    for (int i = 0; i < PX; ++i) {
                        const int q = i * 64 + lane;
                        const float4 val = s_t[(q / PX) * (PX + 1) + (q % PX)];
                        if (p0 + q < HW) o[q] = val;
                    }
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

