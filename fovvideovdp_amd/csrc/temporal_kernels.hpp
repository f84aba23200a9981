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

// "Some sample was outside [0, 1]" (the reference's warning, video_source.py:200), two ways to carry it: a bool -- two
// compares into scalar mask pairs and two scalar ORs per sample -- or, in the register-ring kernels, a running maximum of
// max(V - 1, -V) in ONE vector register (v_add + v_max3 per sample, no scalar registers: those kernels have none to spare).
struct OobMax {
    float m = 0.0f;
};
__device__ __forceinline__ void note_oob(bool& b, float V) { b = b || (V > 1.0f) || (V < 0.0f); }
__device__ __forceinline__ void note_oob(OobMax& b, float V) {
    const float over = V - 1.0f;
    asm("v_max3_f32 %0, %0, %1, -%2" : "+v"(b.m) : "v"(over), "v"(V));
}
__device__ __forceinline__ bool is_oob(bool b) { return b; }
__device__ __forceinline__ bool is_oob(const OobMax& b) { return b.m > 0.0f; }

// Per-channel display model on a float sample V (fvvdp_display_model.py:147-165).  `bad` is set when V was
// outside [0,1] for an EOTF that clamps.
template <int KIND, typename B>
__device__ __forceinline__ float eotf_one(float V, const EotfDev& e, B& bad) {
    if constexpr (KIND == FVVDP_EOTF_SRGB) {
        note_oob(bad, V);
        V = fminf(fmaxf(V, 0.0f), 1.0f);
        // constant divisions as reciprocal multiplies (<= 1 ulp, below the error of the fast log2/exp2 pair)
        const float hi = fast_exp2(2.4f * fast_log2((V + 0.055f) * (1.0f / 1.055f)));
        const float lin = V > 0.04045f ? hi : V * (1.0f / 12.92f);
        return __fadd_rn(__fmul_rn(e.scale, lin), e.y_black);
    } else if constexpr (KIND == FVVDP_EOTF_GAMMA) {
        note_oob(bad, V);
        V = fminf(fmaxf(V, 0.0f), 1.0f);
        const float lin = V > 0.0f ? fast_exp2(e.gamma * fast_log2(V)) : 0.0f;
        return __fadd_rn(__fmul_rn(e.scale, lin), e.y_black);
    } else if constexpr (KIND == FVVDP_EOTF_PQ) {
        note_oob(bad, V);
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

template <typename B>
__device__ __forceinline__ float eotf_f32(float V, const EotfDev& e, B& bad) {
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
// KIND >= 0: the display model is known at compile time (straight-line code, no branch at all).
template <int N, int KIND = -1, typename B = bool>
__device__ __forceinline__ void eotf_apply(float (&V)[N], const EotfDev& e, B& bad) {
    // (round 6: a wave-uniform branch that skips the sRGB toe, as in the YUV kernel, was measured here for 16-bit / float sources:
    // 47.2 -> 46.7 us per 4K frame on uint16 RGB, nothing on float RGB (tools/experiments/r6/s5.sh) -- these kernels wait on memory, not on
    // the toe's three instructions per sample; not kept)
    if constexpr (KIND >= 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) V[i] = eotf_one<KIND>(V[i], e, bad);
        return;
    }
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
            if (e.kind != FVVDP_EOTF_LUT) {           // closed-form display model on code / 65535 (see frame_lum)
#pragma unroll
                for (int i = 0; i < PX; ++i) o[i] = __fmul_rn(eotf_f32((float)code[i] * (1.0f / 65535.0f), e, bad), wc);
            } else {
#pragma unroll
                for (int i = 0; i < PX; ++i) o[i] = __fmul_rn(lut_entry(lut16, code[i], e), wc);
            }
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
    L0Addr out;            // level 0: output frame t of the launch goes to l0_frame(out, t), [HW][4] floats
    int* oob;
    int* ticket;           // nullptr: one workgroup per pixel block; else a zeroed counter -- the grid is the resident capacity and a
                           // workgroup that has finished a block takes the next one (temporal_vec_kernel)
    float taps2[64][2];    // {sustained, transient} tap k (k frames in the past), zero beyond fl: one scalar register pair per tap
    int idx[T_MAX_IDX];    // [FL-1+n_out], entries before the true history are padded with a valid frame
    int idx1[T_MAX_IDX];   // the same for stream 1 (reference): equal to idx for one array per stream, different when the
                           // frames of a stream are separate allocations (fvvdp_temporal_channels_frames)
};

// Entry of a caller-built code-value table.  The range the caller STATED for the table (fvvdp_eotf.L_min / L_max; the host passes
// -FLT_MAX / FLT_MAX when none was stated) is enforced here: the pyramid pass drops clamps on the strength of that range
// (clamps_never_bind in fvvdp_hip.hip), so it must hold whatever the table contains.  A table that keeps its word is unchanged
// (fminf / fmaxf with a bound that does not bind are exact; a NaN entry becomes a bound).
__device__ __forceinline__ float lut_entry(const float* lut, unsigned code, const EotfDev& e) {
    return fminf(fmaxf(lut[code], e.l_min), e.l_max);
}

__device__ __forceinline__ void build_lutw(float* lutw, const EotfDev& e, int C, const float* w, int tid, int nthreads) {
    for (int i = tid; i < 256; i += nthreads) {
        const float l = lut_entry(e.lut, i, e);
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

template <int SRC, int PX, typename FRAME, int KIND = -1, typename B = bool>
__device__ __forceinline__ void frame_lum(const FRAME& f, int C, const float* lutw, const float* lut16,
                                          const float (&w)[3], const EotfDev& e, float (&L)[PX], B& bad) {
    float v[3][PX];
    if constexpr (SRC == SRC_F32) {
        if (C == 3) {
            float t[3 * PX];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < PX; ++i) t[c * PX + i] = f.ch[c].value(i);
            eotf_apply<3 * PX, KIND>(t, e, bad);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < PX; ++i) v[c][i] = __fmul_rn(t[c * PX + i], w[c]);
        } else {
            float t[PX];
#pragma unroll
            for (int i = 0; i < PX; ++i) t[i] = f.ch[0].value(i);
            eotf_apply<PX, KIND>(t, e, bad);
#pragma unroll
            for (int i = 0; i < PX; ++i) v[0][i] = __fmul_rn(t[i], w[0]);
        }
    } else if (SRC == SRC_U16 && (KIND >= 0 ? KIND != FVVDP_EOTF_LUT : e.kind != FVVDP_EOTF_LUT)) {
        // 16-bit codes through the closed-form display model (code / 65535 as the reference unpacks it,
        // video_source.py:186-196): the 65536-entry table lives in global memory and its 6 gathers per pixel made this
        // kernel 2.6x slower than for 8-bit input; same arithmetic as a float source from here on (<= 1e-6 relative)
        if (C == 3) {
            float t[3 * PX];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < PX; ++i) t[c * PX + i] = (float)f.ch[c].code(i) * (1.0f / 65535.0f);
            eotf_apply<3 * PX, KIND>(t, e, bad);
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < PX; ++i) v[c][i] = __fmul_rn(t[c * PX + i], w[c]);
        } else {
            float t[PX];
#pragma unroll
            for (int i = 0; i < PX; ++i) t[i] = (float)f.ch[0].code(i) * (1.0f / 65535.0f);
            eotf_apply<PX, KIND>(t, e, bad);
#pragma unroll
            for (int i = 0; i < PX; ++i) v[0][i] = __fmul_rn(t[i], w[0]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (c > 0 && C != 3) break;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
    #ifdef K1_ABLATE_LUT
                if constexpr (SRC == SRC_U8) v[c][i] = (float)f.ch[c].code(i);
#else
                if constexpr (SRC == SRC_U8) v[c][i] = lutw[c * 256 + f.ch[c].code(i)];
#endif
                else v[c][i] = __fmul_rn(lut_entry(lut16, f.ch[c].code(i), e), w[c]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) L[i] = (C == 3) ? __fadd_rn(__fadd_rn(v[0][i], v[1][i]), v[2][i]) : v[0][i];
}

// ---- closed-form display model on (test, reference) PAIRS (16-bit codes / 65535 and float samples in the register-ring kernels) ----
// eotf_one above, on both streams at once: the same operations in the same order on every component, with the affine steps as packed
// instructions (v_pk_add / v_pk_mul / v_pk_fma), the clamps as one v_med3 per component (fminf(fmaxf()) costs three) and the selects on
// V > 0 / r > 0 dropped where the transcendental pair yields the selected value by itself (k * log2(+0) = -inf, exp2(-inf) = +0 for
// k > 0).  Per sample: PQ 22 -> 11.5 vector instructions (5 of them transcendental), sRGB 13 -> 7.  Rounding: every product and sum is
// rounded on its own, as the reference's torch ops are (fvvdp_display_model.py:147-165, video_source.py:206); in eotf_one the compiler
// fuses `scale * lin + black` and the luminance sums into multiply-adds (HIP's __fmul_rn / __fadd_rn are plain operators), so the two forms
// differ in the last bit of a luminance.  CHECK: flag and clip samples outside [0,1] (float sources; codes / 65535 cannot leave the range).
// No inline assembly here on purpose: an asm instruction that reads the result of v_exp / v_log / v_rcp misses the wait state the compiler
// inserts between a transcendental and its consumer (gfx940+), and reads a stale register now and then.
#ifndef K1_PAIRS
#define K1_PAIRS 1
#endif
template <int N, int KIND, bool CHECK, typename B>
__device__ __forceinline__ void eotf_pairs_exact(v2f (&V)[N], const EotfDev& e, B& bad) {
#pragma clang fp contract(off)       // every product and sum rounded on its own
    if constexpr (CHECK && (KIND == FVVDP_EOTF_SRGB || KIND == FVVDP_EOTF_GAMMA || KIND == FVVDP_EOTF_PQ)) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            note_oob(bad, V[i].x);
            note_oob(bad, V[i].y);
            V[i] = clamp2(V[i], 0.0f, 1.0f);
        }
    }
    if constexpr (KIND == FVVDP_EOTF_SRGB) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const v2f t = (V[i] + splat(0.055f)) * splat(1.0f / 1.055f);
            const v2f ex = v2f{fast_log2(t.x), fast_log2(t.y)} * splat(2.4f);
            const v2f lo = V[i] * splat(1.0f / 12.92f);
            const v2f lin = v2f{V[i].x > 0.04045f ? fast_exp2(ex.x) : lo.x, V[i].y > 0.04045f ? fast_exp2(ex.y) : lo.y};
            V[i] = lin * splat(e.scale) + splat(e.y_black);
        }
    } else if constexpr (KIND == FVVDP_EOTF_GAMMA) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const v2f ex = v2f{fast_log2(V[i].x), fast_log2(V[i].y)} * splat(e.gamma);
            // V = 0: gamma * -inf = -inf, exp2 -> +0 (gamma > 0): the value eotf_one selects
            V[i] = v2f{fast_exp2(ex.x), fast_exp2(ex.y)} * splat(e.scale) + splat(e.y_black);
        }
    } else if constexpr (KIND == FVVDP_EOTF_PQ) {
        const float m = 78.843750000000000f, n = 0.15930175781250000f;
        const float c1 = 0.83593750000000000f, c2 = 18.851562500000000f, c3 = 18.687500000000000f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const v2f lg = v2f{fast_log2(V[i].x), fast_log2(V[i].y)} * splat(1.0f / m);
            const v2f im = v2f{fast_exp2(lg.x), fast_exp2(lg.y)};                        // V = 0 -> +0
            const v2f num = clamp2(im + splat(-c1), 0.0f, 1.0f);                          // max(im - c1, 0): im <= 1, so [0,1] is the same clip
            const v2f den = __builtin_elementwise_fma(im, splat(-c3), splat(c2));         // c2 - c3 * im, one rounding (as contracted in eotf_one)
            const v2f r = num * v2f{fast_rcp(den.x), fast_rcp(den.y)};
            const v2f lr = v2f{fast_log2(r.x), fast_log2(r.y)} * splat(1.0f / n);
            const v2f L = splat(10000.0f) * v2f{fast_exp2(lr.x), fast_exp2(lr.y)};        // r = 0 -> +0
            V[i] = clamp2(L, 0.005f, e.y_peak) + splat(e.y_black);
        }
    } else if constexpr (KIND == FVVDP_EOTF_LINEAR) {
#pragma unroll
        for (int i = 0; i < N; ++i) V[i] = clamp2(V[i], 0.005f, e.y_peak) + splat(e.y_black);
    } else if constexpr (KIND == FVVDP_EOTF_ABSOLUTE) {
#pragma unroll
        for (int i = 0; i < N; ++i) V[i] = clamp2(V[i], e.l_min, e.l_max);
    }
}

// (test, reference) luminance pairs of PX pixels from the raw samples of both streams: frame_lum for closed-form sources, pair by pair
template <int SRC, int PX, int CC, int KIND, typename FRAME, typename B>
__device__ __forceinline__ void frame_lum_pairs(const FRAME& f0, const FRAME& f1, const float (&w)[3], const EotfDev& e, v2f (&L)[PX], B& bad) {
#pragma clang fp contract(off)       // (Lr*w0 + Lg*w1) + Lb*w2 with every product and sum rounded, as frame_lum and video_source.py:206
    static_assert(SRC == SRC_F32 || SRC == SRC_U16, "closed-form sources");
    v2f V[CC * PX];
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            if constexpr (SRC == SRC_F32) {
                // (the two samples are pinned as scalars: left to itself the compiler interleaves the two loaded register quads into
                // pairs with vector shuffles that it lowers THROUGH SCRATCH MEMORY -- 112 bytes of stack, 40 scratch accesses per frame)
                float t0 = f0.ch[c].value(i), t1 = f1.ch[c].value(i);
                asm volatile("" : "+v"(t0), "+v"(t1));
                V[c * PX + i] = v2f{t0, t1};
            }
            else V[c * PX + i] = v2f{(float)f0.ch[c].code(i), (float)f1.ch[c].code(i)} * splat(1.0f / 65535.0f);
        }
    eotf_pairs_exact<CC * PX, KIND, SRC == SRC_F32>(V, e, bad);
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        if constexpr (CC == 3) L[i] = (V[i] * splat(w[0]) + V[PX + i] * splat(w[1])) + V[2 * PX + i] * splat(w[2]);
        else L[i] = V[i] * splat(w[0]);
    }
}

// One filter tap on one pixel: accS += x * f.x, accT += x * f.y for the (test, reference) pair x.  The tap pair f = {sustained,
// transient} sits in ONE scalar register pair and is broadcast by the operand selects of the packed instruction.  Written out
// because the compiler materialises a splat {f, f} pair per tap and channel instead: 4 scalar registers per tap, 256 for a
// 64-tap filter -- they spilled into vector-register lanes and every multiply-add came with ~1.4 v_readlane (r2: 777-4619
// spilled SGPRs in the 32- and 64-slot rings).
__device__ __forceinline__ void fir_tap(v2f& accS, v2f& accT, v2f x, v2f f) {
    asm("v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : "+v"(accS), "+v"(accT) : "v"(x), "s"(f));
}
// The first tap of a sum: accS = x * f.x, accT = x * f.y -- the bits of fir_tap on zeroed accumulators (round(x*f + 0) = round(x*f)),
// without the 2*PX register clears per frame.
__device__ __forceinline__ void fir_tap_first(v2f& accS, v2f& accT, v2f x, v2f f) {
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,1]"
        : "=&v"(accS), "=&v"(accT) : "v"(x), "s"(f));
}
// TAPC taps = 2*TAPC floats of TemporalArgs::taps2 / YuvArgs::taps2, read from the kernel-argument segment with one scalar load
#ifndef K1_TAPC
#define K1_TAPC 4
#endif
constexpr int TAPC = K1_TAPC;
typedef float vtapf __attribute__((ext_vector_type(2 * K1_TAPC)));
typedef vtapf vtapf_a4 __attribute__((aligned(4)));
typedef const vtapf_a4 __attribute__((address_space(4)))* karg_taps_p;

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
        build_lutw(lutw, a.e, a.C, a.w, threadIdx.x, 256);
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
    OobMax bad;
    // .x = test, .y = reference (one register pair per ring value, one v_pk_fma per tap and channel: fir_tap).  Not initialised:
    // every slot is written before its first read (history: slots 0..FL-2, first output step: slot FL-1).
    v2f ring[FL][PX];
    // virtual time v = 0 .. FL-2 is the history, v = FL-1+t the newest frame of output t; ring slot = v % FL.
    // The same pipelined loop fills the history and produces the outputs, so at most one frame is in flight.
    const int total = FL - 1 + a.n_out;
    // window index lists and taps straight from the kernel-argument segment (scalar loads), as in temporal_vec_body
    typedef const int __attribute__((address_space(4)))* karg_int_p;
    typedef const char __attribute__((address_space(4)))* karg_p;
    const karg_p ka = (karg_p)__builtin_amdgcn_kernarg_segment_ptr();
    const karg_int_p idx0 = (karg_int_p)(ka + offsetof(TemporalArgs, idx));
    const karg_int_p idx1 = (karg_int_p)(ka + offsetof(TemporalArgs, idx1));
    RawFrame<SRC, PX> nx[TDIST][2];           // raw samples of the next TDIST frames, in flight
#pragma unroll
    for (int d = 0; d < TDIST; ++d) {
        const size_t off = (size_t)idx0[d < total ? d : total - 1] * a.frame_stride;
        const size_t off1 = (size_t)idx1[d < total ? d : total - 1] * a.frame_stride;
        nx[d][0] = fetch_frame<SRC, PX>(a.src[0], off, a.chan_stride, a.C, px);
        nx[d][1] = fetch_frame<SRC, PX>(a.src[1], off1, a.chan_stride, a.C, px);
    }
    for (int v0 = 0; v0 < total; v0 += FL) {
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int v = v0 + u;
            if (v < total) {
                const RawFrame<SRC, PX> cur0 = nx[u % TDIST][0], cur1 = nx[u % TDIST][1];
                if (v + TDIST < total) {
                    const size_t off = (size_t)idx0[v + TDIST] * a.frame_stride;
                    const size_t off1 = (size_t)idx1[v + TDIST] * a.frame_stride;
                    nx[u % TDIST][0] = fetch_frame<SRC, PX>(a.src[0], off, a.chan_stride, a.C, px);
                    nx[u % TDIST][1] = fetch_frame<SRC, PX>(a.src[1], off1, a.chan_stride, a.C, px);
                }
                {
                    float L0[PX], L1[PX];
                    frame_lum<SRC, PX, RawFrame<SRC, PX>>(cur0, a.C, lutw, a.e.lut, w, a.e, L0, bad);
                    frame_lum<SRC, PX, RawFrame<SRC, PX>>(cur1, a.C, lutw, a.e.lut, w, a.e, L1, bad);
#pragma unroll
                    for (int i = 0; i < PX; ++i) {
                        ring[u][i] = v2f{L0[i], L1[i]};
                        asm volatile("" : "+v"(ring[u][i]));       // pinned where it is produced (see temporal_vec_body::push)
                    }
                }
                if (v >= FL - 1) {
                    v2f accS[PX], accT[PX];                  // (test, reference) of the sustained / the transient channel
#pragma unroll
                    for (int i = 0; i < PX; ++i) accS[i] = accT[i] = v2f{0.0f, 0.0f};
                    // oldest tap first, like the reference's sum over the window dimension.  TAPC taps per scalar load, re-read
                    // in every step through a laundered pointer: hoisted out of the frame loop (or loaded all at once), 2*FL
                    // scalar values stay alive and spill into vector-register lanes (round 3: 1210-1241 spilled SGPRs at FL = 32)
                    karg_p tp = ka + offsetof(TemporalArgs, taps2);
                    asm volatile("" : "+s"(tp));
#pragma unroll
                    for (int c = FL / TAPC - 1; c >= 0; --c) {
                        const vtapf tc = *(karg_taps_p)(tp + c * (8 * TAPC));
#pragma unroll
                        for (int kk = TAPC - 1; kk >= 0; --kk) {
                            const int k = c * TAPC + kk;
                            const int sl = (u - k + 2 * FL) % FL;
                            const v2f f = v2f{tc[2 * kk], tc[2 * kk + 1]};
#pragma unroll
                            for (int i = 0; i < PX; ++i) fir_tap(accS[i], accT[i], ring[sl][i], f);
                        }
                        if constexpr (FL > 16) __builtin_amdgcn_sched_barrier(0);     // one chunk of taps in scalar registers at a time
                    }
                    int t_opaque = v - (FL - 1);              // laundered: no per-step 64-bit offsets pre-computed outside the loop
                    asm volatile("" : "+s"(t_opaque));
                    float* o = l0_frame(a.out, t_opaque);
#pragma unroll
                    for (int i = 0; i < PX; ++i)
                        if (ok[i])
                            *reinterpret_cast<float4*>(o + (size_t)px[i] * 4) = make_float4(accS[i].x, accS[i].y, accT[i].x, accT[i].y);
                }
            }
        }
    }
    if (is_oob(bad) && a.oob) atomicOr(a.oob, 1);
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
    // every source sample is read exactly once by exactly one lane: non-temporal loads (measured -2 % on K1 at 4K)
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    if constexpr (BYTES == 16) {
        const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(q));
        r.wd[0] = t.x; r.wd[1] = t.y; r.wd[2] = t.z; r.wd[3] = t.w;
    } else if constexpr (BYTES == 8) {
        const u2v t = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(q));
        r.wd[0] = t.x; r.wd[1] = t.y;
    } else if constexpr (BYTES == 4) {
        r.wd[0] = __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(q));
    } else if constexpr (BYTES == 2) {
        r.wd[0] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(q));
    } else {
        r.wd[0] = __builtin_nontemporal_load(reinterpret_cast<const unsigned char*>(q));
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

// The loop body is straight-line code (no branch, no workgroup barrier), so that the compiler can count outstanding memory
// operations (s_waitcnt vmcnt(N)) instead of draining them (vmcnt(0)) in every step: with the drain, each step waited for
// the previous step's stores AND its own prefetch, and the long rings (3-4 waves per SIMD) ran latency-bound --
// 4K: 46 -> 34 us/frame at 60 fps, 74 -> 44 us/frame at 120 fps; the 8-slot ring was and stays at the HBM mix ceiling.
// What that takes: colour-channel count and display model as template constants, stores through a buffer resource with
// out-of-range offsets for pixels past the frame, prefetches clamped to the last frame, wave-level LDS ordering, the
// argument block read from the kernel-argument segment, history frames requested in batches.
// TD = frames of raw samples in flight per lane (prefetch distance): 1; 2 measured the same (16-slot) or slower (32-slot).
// The long rings are register-bound (2*FL*PX ring registers): PX = 2 there (PX = 4 on the 16-slot ring: 59 us/frame).
// One wave per block: LDS traffic of the transposes only needs program order (a wave's DS instructions execute in
// order), not a workgroup barrier.  __syncthreads() would also put a release/acquire fence pair around s_barrier, i.e.
// s_waitcnt vmcnt(0): every step would wait for the previous step's stores to be acknowledged and for its own prefetch.
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// CC = number of colour channels as a compile-time constant (3 or 1): with a run-time `C == 3` around the loads and the
// table look-ups, every step has control-flow joins and the compiler falls back to s_waitcnt vmcnt(0).
#ifndef K1_STORE_AUX
#define K1_STORE_AUX 2     // cache policy of the level-0 stores: 2 = nt.  0 / 1 (sc0) / 16 (sc1) / 18: this kernel within its per-process scatter,
                           // but the pyramid pass that follows is 2 % slower behind stores that are not nt (tools/experiments/r4_session44.sh)
#endif
#ifndef K1_EARLY_WORDS
#define K1_EARLY_WORDS 1   // widest sample vector (dwords per channel and lane) whose NEXT frame is requested before the current one is converted
#endif
template <int FL, int PX, int SRC, int TD, int CC, int KIND>
__device__ __forceinline__ void temporal_vec_body(const TemporalArgs& a, const float* lutw, float4* s_t, const int block = (int)blockIdx.x) {
    // the window index lists are the only dynamically indexed members of the argument block: read them straight from the
    // kernel-argument segment (scalar loads), otherwise the whole 2.9 KB block can end up copied to scratch
    typedef const int __attribute__((address_space(4)))* karg_int_p;
    typedef const char __attribute__((address_space(4)))* karg_p;
    const karg_p ka = (karg_p)__builtin_amdgcn_kernarg_segment_ptr();
    const karg_int_p idx0 = (karg_int_p)(ka + offsetof(TemporalArgs, idx));
    const karg_int_p idx1 = (karg_int_p)(ka + offsetof(TemporalArgs, idx1));
    const int lane = threadIdx.x & 63;              // (workgroups of several waves: each wave walks its own pixel block)
    const int p0 = block * (64 * PX);               // first pixel of this wave
    const int pl = min(p0 + lane * PX, a.HW - PX);  // this lane's PX consecutive pixels (clamped: loads stay in range)
    const float w[3] = {CC == 3 ? a.w[0] : 1.0f, a.w[1], a.w[2]};
    OobMax bad;
    // .x = test, .y = reference: one register pair, one v_pk_fma per tap.  Not initialised: every slot is written before
    // its first read (history: slots 0..FL-2, first output step: slot FL-1), so a slot costs registers only once filled.
    v2f ring[FL][PX];
    // virtual time v = 0 .. FL-2 is the history, v = FL-1+t the newest frame of output t; ring slot = v % FL.
    const int total = FL - 1 + a.n_out;
    static_assert(FL % TD == 0, "prefetch slots are indexed with the unrolled ring position");
    RawVecFrame<SRC, PX> nx[TD][2];
    // The loop bodies below are straight-line code: prefetches past the end re-read the last frame (unused), stores of
    // pixels past the frame get an out-of-range buffer offset.  Behind branches the compiler waits for vmcnt(0).
    auto prefetch = [&](int v, RawVecFrame<SRC, PX>& f0, RawVecFrame<SRC, PX>& f1) {
        const int vv = min(v, total - 1);
        f0 = fetch_vec<SRC, PX>(a.src[0], (size_t)idx0[vv] * a.frame_stride + pl, a.chan_stride, CC);
        f1 = fetch_vec<SRC, PX>(a.src[1], (size_t)idx1[vv] * a.frame_stride + pl, a.chan_stride, CC);
    };
    // newest frame -> ring slot u.  The pair is pinned where it is produced: the compiler otherwise sinks the channel sums
    // to their first use (many steps later), keeping three table values per pixel alive instead of one luminance.
    // closed-form display models run on (test, reference) pairs (packed instructions; the reference's roundings)
    constexpr bool PAIRS = (K1_PAIRS != 0) && (SRC == SRC_F32 || SRC == SRC_U16) && KIND >= 0 && KIND != FVVDP_EOTF_LUT;
    auto push = [&](const RawVecFrame<SRC, PX>& c0, const RawVecFrame<SRC, PX>& c1, v2f (&slot)[PX]) {
        if constexpr (PAIRS) {
            frame_lum_pairs<SRC, PX, CC, KIND>(c0, c1, w, a.e, slot, bad);
#pragma unroll
            for (int i = 0; i < PX; ++i) asm volatile("" : "+v"(slot[i]));
        } else {
            float L0[PX], L1[PX];
            frame_lum<SRC, PX, RawVecFrame<SRC, PX>, KIND>(c0, CC, lutw, a.e.lut, w, a.e, L0, bad);
            frame_lum<SRC, PX, RawVecFrame<SRC, PX>, KIND>(c1, CC, lutw, a.e.lut, w, a.e, L1, bad);
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                slot[i] = v2f{L0[i], L1[i]};
                asm volatile("" : "+v"(slot[i]));
            }
        }
    };
    // History (no output yet, almost no arithmetic): HB frames are requested at once, otherwise this phase is one memory
    // latency per frame -- a third of all frames at 120 fps.  The first output frames are requested with the last batch.
    // registers of raw samples per batch: what the ring leaves free when it is nearly full (the last batches)
    constexpr int HB_REGS = FL == 64 ? 16 : ((FL == 32 && RawVec<SRC, PX>::WORDS > 1) ? 24 : 48);
    constexpr int HB = (HB_REGS / (6 * RawVec<SRC, PX>::WORDS)) > 0 ? (HB_REGS / (6 * RawVec<SRC, PX>::WORDS)) : 1;
#pragma unroll
    for (int u0 = 0; u0 < FL - 1; u0 += HB) {
        RawVecFrame<SRC, PX> h[HB][2];
#pragma unroll
        for (int d = 0; d < HB; ++d)
            if (u0 + d < FL - 1) prefetch(u0 + d, h[d][0], h[d][1]);
        if (u0 + HB >= FL - 1) {
#pragma unroll
            for (int d = 0; d < TD; ++d) prefetch(FL - 1 + d, nx[(FL - 1 + d) % TD][0], nx[(FL - 1 + d) % TD][1]);
        }
#pragma unroll
        for (int d = 0; d < HB; ++d)
            if (u0 + d < FL - 1) push(h[d][0], h[d][1], ring[u0 + d]);
        __builtin_amdgcn_sched_barrier(0);          // keep the batches apart: hoisting across them costs registers (spills)
    }
    unsigned int soff[PX];                          // byte offset of this lane's i-th store inside an output frame
#pragma unroll
    for (int i = 0; i < PX; ++i) soff[i] = (p0 + i * 64 + lane < a.HW) ? (unsigned int)(p0 + i * 64 + lane) * 16u : FVVDP_NO_STORE;
    const unsigned int frame_bytes = (unsigned int)a.HW * 16u;      // <= 531 MB (8K)
#ifndef K1_NO_ENTRY_DRAIN
    // One drain on the way in: the loop header joins the history (whose last batch requested the first output frames: loads pending, the
    // first step's registers among the youngest) and the back edge (a step's stores behind the prefetch).  Served by one wait, that
    // came out as vmcnt(1) / vmcnt(0) at the top of every FL frames -- the stores just issued and the prefetch drained.  With nothing
    // pending on the entry path the wait at the header is the back edge's own: counted, stores in flight.
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0) (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)
#endif
    for (int t0 = 0; t0 < a.n_out; t0 += FL) {
#pragma unroll
        for (int j = 0; j < FL; ++j) {
            const int t = t0 + j;
            if (t >= a.n_out) break;
            const int u = (FL - 1 + j) % FL;         // ring slot of the newest frame (compile-time after unrolling)
            // The next frame is requested before this one is converted when a frame is a few registers (8-bit samples); wide
            // samples (12-24 registers per frame pair) are requested after the conversion has freed the registers.
            if constexpr (RawVec<SRC, PX>::WORDS <= K1_EARLY_WORDS) {
                const RawVecFrame<SRC, PX> cur0 = nx[u % TD][0], cur1 = nx[u % TD][1];
                prefetch(FL - 1 + t + TD, nx[u % TD][0], nx[u % TD][1]);
                push(cur0, cur1, ring[u]);
            } else {
                push(nx[u % TD][0], nx[u % TD][1], ring[u]);
                __builtin_amdgcn_sched_barrier(0);
                prefetch(FL - 1 + t + TD, nx[u % TD][0], nx[u % TD][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            v2f accS[PX], accT[PX];                  // (test, reference) of the sustained / the transient channel: set by the first tap
            bool first_tap = true;                   // compile-time after unrolling
            // Taps: TAPC at a time from the kernel-argument segment (scalar cache), oldest first like the reference's sum over the
            // window.  Long filters reload them in every step -- the pointer is laundered so that the loads cannot be hoisted out
            // of the frame loop, where 2*FL scalar values would have to stay alive next to everything else (spills).
            karg_p tp = ka + offsetof(TemporalArgs, taps2);
            if constexpr (FL > 16) asm volatile("" : "+s"(tp));
#ifdef K1_ABLATE_FIR
#pragma unroll
            for (int c = 0; c >= 0; --c) {
#else
#pragma unroll
            for (int c = FL / TAPC - 1; c >= 0; --c) {
#endif
                const vtapf tc = *(karg_taps_p)(tp + c * (8 * TAPC));
#pragma unroll
                for (int kk = TAPC - 1; kk >= 0; --kk) {
                    const int k = c * TAPC + kk;
                    const int sl = (u - k + 2 * FL) % FL;
                    const v2f f = v2f{tc[2 * kk], tc[2 * kk + 1]};
#pragma unroll
                    for (int i = 0; i < PX; ++i) {
                        if (first_tap) fir_tap_first(accS[i], accT[i], ring[sl][i], f);
                        else fir_tap(accS[i], accT[i], ring[sl][i], f);
                    }
                    first_tap = false;
                }
            }
            // transpose through LDS: lane l holds pixels l*PX..l*PX+PX-1, store i writes pixels i*64+l
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < PX; ++i)
                s_t[lane * (PX + 1) + i] = make_float4(accS[i].x, accS[i].y, accT[i].x, accT[i].y);
            wave_lds_order();
            // the frame number is laundered: otherwise the FL per-step offsets j * HW * 16 (64 bit each) are pre-computed outside
            // the loop and, in the long rings, spilled
            int t_opaque = t;
            asm volatile("" : "+s"(t_opaque));
            const __amdgpu_buffer_rsrc_t o = level_rsrc(l0_frame(a.out, t_opaque), frame_bytes);
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const int q = i * 64 + lane;
                const float4 val = s_t[(q / PX) * (PX + 1) + (q % PX)];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{val.x, val.y, val.z, val.w}), o, soff[i], 0, K1_STORE_AUX);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (is_oob(bad) && a.oob) atomicOr(a.oob, 1);
}

template <int FL, int PX, int SRC, int TD, int KIND>
__device__ __forceinline__ void temporal_vec_cc(const TemporalArgs& a, const float* lutw, float4* s_t, const int block = (int)blockIdx.x) {
    if (a.C == 3) temporal_vec_body<FL, PX, SRC, TD, 3, KIND>(a, lutw, s_t, block);
    else temporal_vec_body<FL, PX, SRC, TD, 1, KIND>(a, lutw, s_t, block);
}

// waves per SIMD the register allocation aims at (uint8: 116 / 128 / 168 VGPRs for the 8 / 16 / 32-slot ring)
#ifndef K1_WAVES8
#define K1_WAVES8 4
#endif
#ifndef K1_WAVES16
#define K1_WAVES16 4
#endif
#ifndef K1_WAVES32
#define K1_WAVES32 3
#endif
#ifndef K1_WAVES64
#define K1_WAVES64 3     // 64-slot ring (uint8 only, 1 pixel per lane): above 128 fps
#endif
#ifndef K1_WAVESX8
#define K1_WAVESX8 3     // uint16 / float sources: wider raw samples in flight, closed-form display model
#endif
#ifndef K1_WAVESF8
#define K1_WAVESF8 3     // float source, 8-slot ring
#endif
#ifndef K1_WAVESX16
#define K1_WAVESX16 4
#endif
#ifndef K1_WAVESX32
#define K1_WAVESX32 2
#endif
constexpr int k1_waves(int FL, int SRC) {
    return SRC == SRC_U8 ? (FL == 8 ? K1_WAVES8 : (FL == 16 ? K1_WAVES16 : (FL == 32 ? K1_WAVES32 : K1_WAVES64)))
                         : FL == 64 ? K1_WAVES64 : (FL == 8 ? (SRC == SRC_F32 ? K1_WAVESF8 : K1_WAVESX8) : (FL == 16 ? K1_WAVESX16 : K1_WAVESX32));
}
#ifdef K1_TIMELINE         // profiling build: (start, end) of every workgroup on the 100 MHz wall clock (tools/gpu_timeline.py k1)
static __device__ unsigned long long g_k1_timeline[4 * 65536];
__device__ __forceinline__ void k1_clock_record(int block, unsigned long long t0) {
    if (threadIdx.x == 0 && block < 65536) {
        unsigned long long* t = g_k1_timeline + 4 * (size_t)block;
        t[0] = t0;
        t[1] = wall_clock64();
        t[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) << 32) | (unsigned int)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);
        t[3] = (unsigned long long)block;
    }
}
#endif
// Waves per workgroup: the 8-slot ring runs K1_WPB8 = 4 waves per workgroup, each on its own pixel block, the four blocks adjacent
// in memory (1 KiB of every source plane and 16 KiB of level 0 per frame and workgroup instead of 256 B / 4 KiB).  The waves share
// the code-value table and nothing else (no barrier after the table is built); a workgroup's waves start together and stay close,
// so the memory system sees longer contiguous runs: -1.3 ... -1.5 us per 4K frame on every destination buffer, slow or fast
// (tools/microbench/k1_stream.hip "k1_4w" against "k1", profiles/r05_k1_mode.md; 8 waves: the same; 2 waves: half of it).
#ifndef K1_WPB8
#define K1_WPB8 4
#endif
constexpr int k1_wpb(int FL) { return FL == 8 ? K1_WPB8 : 1; }
template <int FL, int PX, int SRC, int TD = 1>
__global__ __launch_bounds__(64 * k1_wpb(FL), k1_waves(FL, SRC))
void temporal_vec_kernel(const TemporalArgs a_byval) {

    // All reads of the argument block go to the kernel-argument segment itself (scalar loads).  Through the by-value
    // parameter the compiler starts from a private copy and, in the largest instantiations, fails to remove it: 2.9 KB of
    // scratch per lane and every filter tap reloaded from it.
    const TemporalArgs& a = *(const TemporalArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_byval;
    constexpr int WPB = k1_wpb(FL);
    __shared__ float lutw[SRC == SRC_U8 ? 768 : 1];
    __shared__ float4 s_t_all[WPB][64 * (PX + 1)];  // per wave: one padded row of PX float4 per lane
    if constexpr (SRC == SRC_U8) build_lutw(lutw, a.e, a.C, a.w, threadIdx.x, 64 * WPB);
    __syncthreads();
    const int wave = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    float4* const s_t = s_t_all[wave];
    const int block0 = (int)blockIdx.x * WPB + wave;          // this wave's pixel block
    if constexpr (WPB > 1) {
        if (block0 * (64 * PX) >= a.HW) return;                // (the last workgroup of a frame whose block count is not a multiple)
    }
    // colour-channel count and display model as compile-time constants of the loop body (see temporal_vec_body)
    if constexpr (SRC == SRC_U8 && FL == 16) {
        // uint8 at 33-64 fps: with a ticket counter the grid is the resident capacity and a workgroup that has finished a block of
        // pixels takes the next one -- the XCDs of a box run this memory-bound kernel 5-8 % apart, and the hardware hands each of
        // them exactly an eighth of the workgroups (profiles/r04_lockstep.md, section 5: -2 ... -6 % with the 16-slot ring; no gain
        // with the 8-slot ring, which takes the plain path below)
        typedef const char __attribute__((address_space(4)))* karg_p;
        int block = block0;
        const int n_blocks = (a.HW + 64 * PX - 1) / (64 * PX);
        for (;;) {
            int next = 0x7fffffff;
            {
                karg_p ka = (karg_p)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(ka));
                int* const tk = *(int* const __attribute__((address_space(4)))*)(ka + offsetof(TemporalArgs, ticket));
                if (tk && threadIdx.x == 0) next = (int)gridDim.x + atomicAdd(tk, 1);       // asked for now, looked at after the block
            }
#ifdef K1_TIMELINE
            const unsigned long long k1_t0 = wall_clock64();
#endif
            temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_LUT>(a, lutw, s_t, block);
#ifdef K1_TIMELINE
            k1_clock_record(block, k1_t0);
#endif
            block = __builtin_amdgcn_readfirstlane(next);
            if (block >= n_blocks) break;
            wave_lds_order();                        // the transposes of the next block reuse s_t
        }
    } else if constexpr (SRC == SRC_U8) {
        temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_LUT>(a, lutw, s_t, block0);
    } else if constexpr (FL == 64) {
        // 64-slot ring for 16-bit / float input (k1_ring64_ok() is the host-side list): RGB behind an sRGB or PQ display,
        // and float luminance frames (what custom video sources deliver).  Everything else above 128 fps: generic kernel.
        if (a.C == 3 && a.e.kind == FVVDP_EOTF_SRGB) temporal_vec_body<FL, PX, SRC, TD, 3, FVVDP_EOTF_SRGB>(a, lutw, s_t, block0);
        else if (a.C == 3 && a.e.kind == FVVDP_EOTF_PQ) temporal_vec_body<FL, PX, SRC, TD, 3, FVVDP_EOTF_PQ>(a, lutw, s_t, block0);
        else if constexpr (SRC == SRC_F32) {
            if (a.C == 1 && a.e.kind == FVVDP_EOTF_NONE) temporal_vec_body<FL, PX, SRC, TD, 1, FVVDP_EOTF_NONE>(a, lutw, s_t, block0);
        }
    } else {
        switch (a.e.kind) {
            case FVVDP_EOTF_SRGB: temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_SRGB>(a, lutw, s_t, block0); break;
            case FVVDP_EOTF_GAMMA: temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_GAMMA>(a, lutw, s_t, block0); break;
            case FVVDP_EOTF_PQ: temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_PQ>(a, lutw, s_t, block0); break;
            case FVVDP_EOTF_LINEAR: temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_LINEAR>(a, lutw, s_t, block0); break;
            case FVVDP_EOTF_ABSOLUTE: temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_ABSOLUTE>(a, lutw, s_t, block0); break;
            case FVVDP_EOTF_LUT:
                if constexpr (SRC == SRC_U16) temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_LUT>(a, lutw, s_t, block0);
                break;                               // (a table for a float source is refused by the host side)
            default: temporal_vec_cc<FL, PX, SRC, TD, FVVDP_EOTF_NONE>(a, lutw, s_t, block0); break;   // luminances already
        }
    }
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
    L0Addr out;               // level 0 (see TemporalArgs::out)
    int* oob;
    float taps2[32][2];     // {sustained, transient} tap k, see TemporalArgs
    int idx[T_MAX_IDX];
};

template <typename T, typename B>
__device__ __forceinline__ float yuv_lum(const T* __restrict__ f, const YuvArgs& a, int p, B& bad) {
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
    OobMax bad;
    v2f ring[FL][PX];               // .x = test, .y = reference; every slot is written before its first read
    const int total = FL - 1 + a.n_out;
    typedef const int __attribute__((address_space(4)))* karg_int_p;
    typedef const char __attribute__((address_space(4)))* karg_p;
    const karg_p ka = (karg_p)__builtin_amdgcn_kernarg_segment_ptr();
    const karg_int_p idx0 = (karg_int_p)(ka + offsetof(YuvArgs, idx));
    for (int v0 = 0; v0 < total; v0 += FL) {
#pragma unroll
        for (int u = 0; u < FL; ++u) {
            const int v = v0 + u;
            if (v < total) {
                const size_t off = (size_t)idx0[v] * a.frame_stride;
                const T* f0 = reinterpret_cast<const T*>(a.src[0]) + off;
                const T* f1 = reinterpret_cast<const T*>(a.src[1]) + off;
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    ring[u][i] = v2f{yuv_lum<T>(f0, a, px[i], bad), yuv_lum<T>(f1, a, px[i], bad)};
                    asm volatile("" : "+v"(ring[u][i]));
                }
                if (v >= FL - 1) {
                    v2f accS[PX], accT[PX];
#pragma unroll
                    for (int i = 0; i < PX; ++i) accS[i] = accT[i] = v2f{0.0f, 0.0f};
                    // taps: TAPC per scalar load, laundered pointer, one chunk alive at a time (see temporal_ring_kernel)
                    karg_p tp = ka + offsetof(YuvArgs, taps2);
                    asm volatile("" : "+s"(tp));
#pragma unroll
                    for (int c = FL / TAPC - 1; c >= 0; --c) {
                        const vtapf tc = *(karg_taps_p)(tp + c * (8 * TAPC));
#pragma unroll
                        for (int kk = TAPC - 1; kk >= 0; --kk) {
                            const int k = c * TAPC + kk;
                            const int sl = (u - k + 2 * FL) % FL;
                            const v2f f = v2f{tc[2 * kk], tc[2 * kk + 1]};
#pragma unroll
                            for (int i = 0; i < PX; ++i) fir_tap(accS[i], accT[i], ring[sl][i], f);
                        }
                        if constexpr (FL > 16) __builtin_amdgcn_sched_barrier(0);
                    }
                    int t_opaque = v - (FL - 1);
                    asm volatile("" : "+s"(t_opaque));
                    float* o = l0_frame(a.out, t_opaque);
#pragma unroll
                    for (int i = 0; i < PX; ++i)
                        if (ok[i])
                            *reinterpret_cast<float4*>(o + (size_t)px[i] * 4) = make_float4(accS[i].x, accS[i].y, accT[i].x, accT[i].y);
                }
            }
        }
    }
    if (is_oob(bad) && a.oob) atomicOr(a.oob, 1);
}

// ---- vector variant of the YUV ingest (the fast path for W % 4 == 0) -----------------------------------------------
// Single-wave workgroups, a lane owns PX CONSECUTIVE pixels of one image row (x0 = PX*j; PX = 4 or 2).  Everything that depends
// only on the pixel position (plane offsets, the bilinear weights of the 4:2:0 chroma upsampling) is computed once per
// lane; per frame a lane issues one Y load (PX samples) and, per chroma plane and source row, one load of its own PX/2 chroma
// columns; the two neighbour columns (clamped) come from the adjacent lanes through DPP
// (5 loads per lane, frame and stream instead of 13, and no edge-lane branches).
// Test and reference stream are converted together as packed (test, reference) pairs (v_pk_* instructions); the
// result agrees with yuv_lum above to rounding order.  Raw samples of the next frame are prefetched while the
// current one is converted, and the finished float4 pixels go through the same LDS transpose as temporal_vec_kernel.
// PX = 2 halves the registers of the window (2*FL*PX) and of everything per pixel: more waves per SIMD where the window is what
// limits them (the 16-slot window above all), at the price of twice the per-lane work (loads, stores, branches) per pixel.
template <typename T, bool C420, int PX>
struct YuvRaw {
    static constexpr int YW = (PX * (int)sizeof(T) + 3) / 4;      // dwords holding PX samples (PX = 2, 8 bit: half a dword)
    unsigned int y[YW];
    // 4:2:0: cp[plane][row] = the lane's own chroma columns (PX = 4: columns 2j, 2j+1 packed; PX = 2: column j); the neighbour
    // columns are the finished values of the adjacent lanes (DPP; lanes 0 and 63 of a wave are halo lanes that only supply them).
    // 4:4:4: cp[plane][0..YW-1] = PX samples.
    unsigned int cp[2][2];
    __device__ __forceinline__ float ysample(int i) const {
        if constexpr (sizeof(T) == 1) return (float)((y[0] >> (8 * i)) & 0xFFu);
        else return (float)((y[i / 2] >> (16 * (i % 2))) & 0xFFFFu);
    }
    __device__ __forceinline__ float c444(int pl, int i) const {
        if constexpr (sizeof(T) == 1) return (float)((cp[pl][0] >> (8 * i)) & 0xFFu);
        else return (float)((cp[pl][i / 2] >> (16 * (i % 2))) & 0xFFFFu);
    }
    static __device__ __forceinline__ float lo(unsigned int w) { return (float)(w & (sizeof(T) == 1 ? 0xFFu : 0xFFFFu)); }
    static __device__ __forceinline__ float hi(unsigned int w) {
        return (float)((w >> (8 * (int)sizeof(T))) & (sizeof(T) == 1 ? 0xFFu : 0xFFFFu));
    }
};

struct YuvGeom {           // per-lane constants
    unsigned int oy;       // BYTE offset of the PX luma samples inside a frame (a frame is at most 3 planes of 2-byte samples: < 4 GiB up to 715 Mpixel)
    unsigned int oc[2][2]; // BYTE offsets inside a frame of the lane's own chroma columns, [plane][source row] (4:4:4: [plane][0])
    bool left_own, right_own;   // the clamped neighbour column is one of the lane's own (image edges)
};

// BYTES consecutive bytes at byte offset `off` of the frame behind the buffer resource `f` (aligned to BYTES) into the low bits of dwords.
// Buffer loads take the frame's address from scalar registers (the resource) and the lane's offset from one 32-bit vector register:
// no vector instruction goes into addressing (global loads: a 64-bit vector add per load, 10 per lane and frame).
template <int BYTES>
__device__ __forceinline__ void yuv_load_bytes(__amdgpu_buffer_rsrc_t f, unsigned int off, unsigned int* out) {
    if constexpr (BYTES == 1) out[0] = (unsigned int)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(f, (int)off, 0, 0);
    else if constexpr (BYTES == 2) out[0] = (unsigned int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(f, (int)off, 0, 0);
    else if constexpr (BYTES == 4) out[0] = (unsigned int)__builtin_amdgcn_raw_buffer_load_b32(f, (int)off, 0, 0);
    else {
        static_assert(BYTES == 8, "1, 2, 4 or 8 bytes");
        const v2i t = __builtin_bit_cast(v2i, __builtin_amdgcn_raw_buffer_load_b64(f, (int)off, 0, 0));
        out[0] = (unsigned int)t.x; out[1] = (unsigned int)t.y;
    }
}

// f = the frame (wave-uniform); every load is the frame + a per-lane constant
template <typename T, bool C420, int PX>
__device__ __forceinline__ YuvRaw<T, C420, PX> yuv_fetch(__amdgpu_buffer_rsrc_t f, const YuvGeom& g) {
    YuvRaw<T, C420, PX> r;
    yuv_load_bytes<PX * (int)sizeof(T)>(f, g.oy, r.y);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        if constexpr (C420) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) yuv_load_bytes<(PX / 2) * (int)sizeof(T)>(f, g.oc[pl][rr], &r.cp[pl][rr]);
        } else {
            yuv_load_bytes<PX * (int)sizeof(T)>(f, g.oc[pl][0], r.cp[pl]);
        }
    }
    return r;
}

// RGB (clipped to [0,1]) of the lane's PX pixels for both streams at once: rgb[3*i+c] = (test, reference).
// 4:2:0 chroma: the source columns are blended vertically first, then horizontally (2 x fewer products than the
// per-pixel form; the result differs from it by rounding order only).
// a * s + c clipped to [0, 1], both components: v_pk_fma_f32 with the clamp modifier
__device__ __forceinline__ v2f pfma_clamp01(v2f a, float s, v2f c) {
    v2f r;
    const v2f sv = splat(s);
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "s"(sv), "v"(c));      // the factor is wave-uniform: scalar register pair
    return r;
}

// STDM: the colour matrix has the shape of every ITU YCbCr matrix -- R = Y + m2 Cr, G = Y + m4 Cb + m5 Cr, B = Y + m7 Cb (unit luma
// column, no Cb in red, no Cr in blue; the host checks the nine numbers, yuv_matrix_is_standard).  Multiplying by 1 and adding 0 * x
// are exact, so the four multiply-adds left per pixel give the same bits as the nine of the general form.
template <typename T, bool C420, bool STDM, int PX>
__device__ __forceinline__ void yuv_pair_rgb(const YuvRaw<T, C420, PX>& r0, const YuvRaw<T, C420, PX>& r1, const YuvArgs& a,
                                             const YuvGeom& g, float fy, float gy, float fx0, float gx0, v2f (&rgb)[3 * PX]) {
    auto cf = [&](float c0, float c1) { return clamp2(pfma(v2f{c0, c1}, a.wc, splat(-(128.0f / 224.0f))), -0.5f, 0.5f); };
    auto from_left = [](v2f v) { return v2f{__uint_as_float(lane_left_u32(__float_as_uint(v.x))), __uint_as_float(lane_left_u32(__float_as_uint(v.y)))}; };
    auto from_right = [](v2f v) { return v2f{__uint_as_float(lane_right_u32(__float_as_uint(v.x))), __uint_as_float(lane_right_u32(__float_as_uint(v.y)))}; };
    v2f uv[2][PX];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        if constexpr (C420) {
            using R = YuvRaw<T, C420, PX>;
            // A lane converts ONLY its own chroma columns of the two source rows and blends them vertically; the neighbour columns
            // are the finished values of the adjacent lanes (same image row, hence the same vertical weights) and arrive through
            // DPP -- half the conversions, clamps and vertical blends of fetching the neighbours' raw words and converting all four
            // columns in every lane (round 6: 160 -> 80 vector instructions per lane and frame for the chroma of 4 pixels x 2
            // streams; the same values, bit for bit).  At a row start / end the clamped neighbour column is one of the lane's own.
            // An even pixel 2c blends column c (weight fx = .75, or 0 at x = 0) with column c-1, an odd pixel 2c+1 column c (.75)
            // with column c+1 (.25) -- torch's bilinear x2, align_corners=False.
            if constexpr (PX == 4) {
                v2f own[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const v2f a0 = k == 0 ? cf(R::lo(r0.cp[pl][0]), R::lo(r1.cp[pl][0])) : cf(R::hi(r0.cp[pl][0]), R::hi(r1.cp[pl][0]));
                    const v2f a1 = k == 0 ? cf(R::lo(r0.cp[pl][1]), R::lo(r1.cp[pl][1])) : cf(R::hi(r0.cp[pl][1]), R::hi(r1.cp[pl][1]));
                    own[k] = pfma(a1, fy, a0 * gy);
                }
                const v2f nl = from_left(own[1]);
                const v2f nr = from_right(own[0]);
                v2f col[4];
                col[0] = g.left_own ? own[0] : nl;
                col[1] = own[0];
                col[2] = own[1];
                col[3] = g.right_own ? own[1] : nr;
                uv[pl][0] = pfma(col[1], fx0, col[0] * gx0);
                uv[pl][1] = pfma(col[2], 0.25f, col[1] * 0.75f);
                uv[pl][2] = pfma(col[2], 0.75f, col[1] * 0.25f);
                uv[pl][3] = pfma(col[3], 0.25f, col[2] * 0.75f);
            } else {
                static_assert(PX == 2, "4 or 2 pixels per lane");
                const v2f a0 = cf(R::lo(r0.cp[pl][0]), R::lo(r1.cp[pl][0]));
                const v2f a1 = cf(R::lo(r0.cp[pl][1]), R::lo(r1.cp[pl][1]));
                const v2f own = pfma(a1, fy, a0 * gy);
                const v2f nl = from_left(own);
                const v2f nr = from_right(own);
                const v2f cl = g.left_own ? own : nl;
                const v2f cr = g.right_own ? own : nr;
                uv[pl][0] = pfma(own, fx0, cl * gx0);
                uv[pl][1] = pfma(cr, 0.25f, own * 0.75f);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PX; ++i) uv[pl][i] = cf(r0.c444(pl, i), r1.c444(pl, i));
        }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        // the [0,1] clips ride on the multiply-add that produces the value (VOP3P clamp bit): one instruction instead of three
        const v2f Yf = pfma_clamp01(v2f{r0.ysample(i), r1.ysample(i)}, a.wy, splat(-(16.0f / 219.0f)));
        if constexpr (STDM) {
            rgb[3 * i + 0] = pfma_clamp01(uv[1][i], a.m[2], Yf);
            rgb[3 * i + 1] = pfma_clamp01(uv[1][i], a.m[5], pfma(uv[0][i], a.m[4], Yf));
            rgb[3 * i + 2] = pfma_clamp01(uv[0][i], a.m[7], Yf);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                v2f v = Yf * a.m[3 * c];
                v = pfma(uv[0][i], a.m[3 * c + 1], v);
                rgb[3 * i + c] = pfma_clamp01(uv[1][i], a.m[3 * c + 2], v);
            }
        }
    }
}

// Display model on N (test, reference) pairs whose values are already inside [0,1]; one wave-uniform branch.
// The affine parts run packed and fused (one multiply-add where fvvdp_display_model.py:160-165 rounds twice): the power itself goes
// through the hardware's log2 / exp2 (relative error up to ~1e-6 on the dark end), an order of magnitude above what a rounding of an
// affine step moves, so mimicking the reference's separate roundings bought nothing here (round 6: -32 of 448 vector instructions per
// lane and frame; the 8-bit RGB path, which is table-exact, is a different kernel).
template <int N, int KIND>
__device__ __forceinline__ void eotf_apply_pairs(v2f (&V)[N], const EotfDev& e) {
    bool bad = false;
    switch (KIND) {                                  // compile-time: one case survives, the loop body stays branch-free
        case FVVDP_EOTF_SRGB: {
            // The linear toe (V <= 0.04045, i.e. 8-bit codes <= 10) costs a compare, a select and half a multiply per value: 5 of 14
            // vector instructions per (test, reference) pair.  One wave-uniform branch on the smallest of the 2N values (N/2 + N/4 + ...
            // v_min3) takes the power branch alone where no lane needs the toe -- the same expression for those values, same bits.
            float mn;
            asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(V[0].x), "v"(V[0].y));
#pragma unroll
            for (int i = 1; i < N; ++i) asm("v_min3_f32 %0, %0, %1, %2" : "+v"(mn) : "v"(V[i].x), "v"(V[i].y));
            if (__builtin_amdgcn_ballot_w64(!(mn > 0.04045f)) == 0) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const v2f t = pfma(V[i], 1.0f / 1.055f, splat(0.055f / 1.055f));
                    const v2f ex = v2f{fast_log2(t.x), fast_log2(t.y)} * 2.4f;
                    V[i] = pfma(v2f{fast_exp2(ex.x), fast_exp2(ex.y)}, e.scale, splat(e.y_black));
                }
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const v2f t = pfma(V[i], 1.0f / 1.055f, splat(0.055f / 1.055f));
                    const v2f ex = v2f{fast_log2(t.x), fast_log2(t.y)} * 2.4f;
                    const v2f lo = V[i] * (1.0f / 12.92f);
                    const v2f lin = v2f{V[i].x > 0.04045f ? fast_exp2(ex.x) : lo.x, V[i].y > 0.04045f ? fast_exp2(ex.y) : lo.y};
                    V[i] = pfma(lin, e.scale, splat(e.y_black));
                }
            }
            break;
        }
        case FVVDP_EOTF_GAMMA: {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const v2f ex = v2f{fast_log2(V[i].x), fast_log2(V[i].y)} * e.gamma;
                const v2f lin = v2f{V[i].x > 0.0f ? fast_exp2(ex.x) : 0.0f, V[i].y > 0.0f ? fast_exp2(ex.y) : 0.0f};
                V[i] = pfma(lin, e.scale, splat(e.y_black));
            }
            break;
        }
        case FVVDP_EOTF_PQ: {
            // the packed form of eotf_one<PQ> (the same values: its one fused step, c2 - c3 * V^(1/m), is fused there as well)
            eotf_pairs_exact<N, FVVDP_EOTF_PQ, false>(V, e, bad);
            break;
        }
        case FVVDP_EOTF_LINEAR: {
#pragma unroll
            for (int i = 0; i < N; ++i) V[i] = clamp2(V[i], 0.005f, e.y_peak) + e.y_black;
            break;
        }
        case FVVDP_EOTF_ABSOLUTE: {
#pragma unroll
            for (int i = 0; i < N; ++i) V[i] = clamp2(V[i], e.l_min, e.l_max);
            break;
        }
        default: break;
    }
}

// luminance R*w0 + G*w1 + B*w2 (video_source.py:206) as one multiply and two multiply-adds (see eotf_apply_pairs)
__device__ __forceinline__ v2f lum_pair(v2f r, v2f g, v2f b, float w0, float w1, float w2) {
    return pfma(b, w2, pfma(g, w1, r * w0));
}

// Window of the last FL luminance pairs of a lane's PX pixels, kept in registers WITHOUT moving it: frame v goes to slot
// v mod FL and the FIR of that frame is one of FL straight-line variants (a wave-uniform switch over v mod FL), each with
// compile-time slots and taps: age k sits in slot (S - k) mod FL.  (Shifting the window by one slot per frame instead cost
// FL*8-8 register moves per frame, a quarter of the 16-slot kernel's VALU instructions.)  Oldest tap first, like the
// reference's sum over the window; two taps per scalar load: this kernel has the colour matrix, the display model and the
// chroma weights in scalar registers next to the taps.
template <int FL, int S, int PX>
__device__ __forceinline__ void yuv_window_step(v2f (&win)[FL][PX], const v2f (&lum)[PX],
                                                const char __attribute__((address_space(4)))* tp, v2f (&acc_s)[PX], v2f (&acc_t)[PX]) {
    typedef float v4tap __attribute__((ext_vector_type(4), aligned(4)));
    typedef const v4tap __attribute__((address_space(4)))* karg_tap4_p;
#pragma unroll
    for (int i = 0; i < PX; ++i) win[S][i] = lum[i];
#pragma unroll
    for (int c = FL / 2 - 1; c >= 0; --c) {
        const v4tap tc = *(karg_tap4_p)(tp + c * 16);
#pragma unroll
        for (int kk = 1; kk >= 0; --kk) {
            const int k = c * 2 + kk;
            const v2f f = v2f{tc[2 * kk], tc[2 * kk + 1]};
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                if (k == FL - 1) fir_tap_first(acc_s[i], acc_t[i], win[(S - k + FL) % FL][i], f);      // oldest tap first: it sets the sums
                else fir_tap(acc_s[i], acc_t[i], win[(S - k + FL) % FL][i], f);
            }
        }
    }
}

template <int FL, int PX>
__device__ __forceinline__ void yuv_window_dispatch(int slot, v2f (&win)[FL][PX], const v2f (&lum)[PX],
                                                    const char __attribute__((address_space(4)))* tp, v2f (&acc_s)[PX], v2f (&acc_t)[PX]) {
    static_assert(FL <= 16, "one case per slot below");
    switch (slot) {
        default: __builtin_unreachable();
#define FVVDP_YUV_CASE(N) case N: if constexpr (N < FL) yuv_window_step<FL, (N < FL ? N : 0), PX>(win, lum, tp, acc_s, acc_t); break;
        FVVDP_YUV_CASE(0) FVVDP_YUV_CASE(1) FVVDP_YUV_CASE(2) FVVDP_YUV_CASE(3) FVVDP_YUV_CASE(4) FVVDP_YUV_CASE(5)
        FVVDP_YUV_CASE(6) FVVDP_YUV_CASE(7) FVVDP_YUV_CASE(8) FVVDP_YUV_CASE(9) FVVDP_YUV_CASE(10) FVVDP_YUV_CASE(11)
        FVVDP_YUV_CASE(12) FVVDP_YUV_CASE(13) FVVDP_YUV_CASE(14) FVVDP_YUV_CASE(15)
#undef FVVDP_YUV_CASE
    }
}

#define YUV_QUADS 62     // pixel groups (PX consecutive pixels, one per lane) written per wave
#ifndef YUV_TD8
#define YUV_TD8 2        // frames of raw samples in flight per lane, 8-slot window
#endif
#ifndef YUV_TD16
#define YUV_TD16 1       // 16-slot window: 128 registers of window leave room for one frame of raw samples (2 -> 2-11 spilled dwords)
#endif
// One rolled loop over the frames, straight-line inside (see temporal_vec_kernel for why: counted waits instead of
// drains).  The conversion of a frame is a few hundred instructions and exists once; the window of the last FL luminance
// pairs stays where it is (yuv_window_step above) and only the short FIR exists FL times.
// The FIR also runs during the history frames (its result is dropped by an out-of-range store offset).
template <int FL, typename T, bool C420, int KIND, bool STDM, int PX>
__device__ __forceinline__ void temporal_yuv_vec_body(const YuvArgs& a, float4* s_t, const int lane, const int block) {
    constexpr int TD = FL <= 8 ? YUV_TD8 : YUV_TD16;
    typedef const int __attribute__((address_space(4)))* karg_int_p;
    typedef const char __attribute__((address_space(4)))* karg_p;
    const karg_int_p idx = (karg_int_p)((karg_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(YuvArgs, idx));
    const int HW = a.W * a.H;
    const int uvplane = a.uvw * a.uvh;
    // lanes 1..62 own the wave's 62 pixel groups; lanes 0 and 63 convert the groups next to them, only to hand their
    // chroma columns to lanes 1 and 62 (quads are clamped to the frame: a clamped lane duplicates its neighbour,
    // which then sits at a row start / end and does not look at it)
    const int p0 = block * (YUV_QUADS * PX);
#ifdef YUV_ABLATE_MEM      // profiling ablation: every wave converts the same 64 quads of source frame 0 (cache hits), nothing is stored
    const int quad = lane;
#else
    const int quad = min(max(block * YUV_QUADS - 1 + lane, 0), HW / PX - 1);
#endif
    const int pl = quad * PX;
    YuvGeom g;
    float g_fy = 0.0f, g_gy = 0.0f;      // vertical bilinear weights
    float g_fx0 = 0.0f, g_gx0 = 0.0f;    // horizontal weights of pixel 0 (0/1 at the left image edge, else .75/.25)
    {
        const int y = pl / a.W, x = pl - y * a.W;
        constexpr unsigned int ES = (unsigned int)sizeof(T);
        g.oy = (unsigned int)pl * ES;
        if constexpr (C420) {
            // torch bilinear, align_corners=False: source = (dst + 0.5)/2 - 0.5 clamped at 0 (video_source_file.py:262-266)
            const float sy = fmaxf(((float)y + 0.5f) * 0.5f - 0.5f, 0.0f);
            const int y0 = (int)sy, y1 = min(y0 + 1, a.uvh - 1);
            g_fy = sy - (float)y0;
            g_gy = 1.0f - g_fy;
            const int j2 = x >> 1;                       // first of the lane's own PX/2 chroma columns
#pragma unroll
            for (int cpl = 0; cpl < 2; ++cpl) {
                g.oc[cpl][0] = (unsigned int)(HW + cpl * uvplane + y0 * a.uvw + j2) * ES;
                g.oc[cpl][1] = (unsigned int)(HW + cpl * uvplane + y1 * a.uvw + j2) * ES;
            }
            // lanes to the left / right hold the adjacent quad of the same row unless this lane starts / ends the row
            // (lanes clamped to the last quad of the frame sit at a row end as well)
            g.left_own = (j2 == 0);
            g.right_own = (j2 + PX / 2 > a.uvw - 1);
            const float sx = fmaxf(((float)x + 0.5f) * 0.5f - 0.5f, 0.0f);
            g_fx0 = sx - (float)(int)sx;                 // 0 at x == 0 (then the "left" column is column 0 itself), else .75
            g_gx0 = 1.0f - g_fx0;
        } else {
            g.oc[0][0] = g.oc[0][1] = (unsigned int)(HW + pl) * ES;
            g.oc[1][0] = g.oc[1][1] = (unsigned int)(HW + uvplane + pl) * ES;
            g.left_own = g.right_own = false;
        }
    }
    v2f win[FL][PX];                                      // (test, reference) luminance; frame v of the window in slot v mod FL
#pragma unroll
    for (int u = 0; u < FL; ++u)
#pragma unroll
        for (int i = 0; i < PX; ++i) win[u][i] = splat(0.0f);
    const int total = FL - 1 + a.n_out;
    const unsigned int src_bytes = (unsigned int)(HW + 2 * uvplane) * (unsigned int)sizeof(T);      // one planar frame
    auto prefetch = [&](int v, YuvRaw<T, C420, PX>& f0, YuvRaw<T, C420, PX>& f1) {
#ifdef YUV_ABLATE_MEM
        const size_t off = (size_t)(idx[min(v, total - 1)] & 0) * a.frame_stride;
#else
        const size_t off = (size_t)idx[min(v, total - 1)] * a.frame_stride;      // past the end: the last frame again (unused)
#endif
        f0 = yuv_fetch<T, C420, PX>(src_rsrc(reinterpret_cast<const char*>(a.src[0]) + off * sizeof(T), src_bytes), g);
        f1 = yuv_fetch<T, C420, PX>(src_rsrc(reinterpret_cast<const char*>(a.src[1]) + off * sizeof(T), src_bytes), g);
    };
    YuvRaw<T, C420, PX> nx[TD][2];
#pragma unroll
    for (int d = 0; d < TD; ++d) prefetch(d, nx[d][0], nx[d][1]);
    // store i of the lane writes pixel i*64+lane of the wave's run of 62*PX pixels (row `lane` of the LDS tile = group lane-1)
    unsigned int soff[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int q = i * 64 + lane;
        soff[i] = (q < YUV_QUADS * PX && p0 + q < HW) ? (unsigned int)(p0 + q) * 16u : FVVDP_NO_STORE;
    }
    const unsigned int frame_bytes = (unsigned int)HW * 16u;
#ifndef YUV_NO_ENTRY_DRAIN
    // The loop header joins the entry path (TD frames of loads pending, the registers the first step reads among the LAST requested) and
    // the back edge (the same registers requested two steps ago, with a step's stores and the other slot's loads behind them): the
    // compiler's wait at the top of the loop has to serve both and came out as vmcnt(1) / vmcnt(0) -- a drain of the stores just issued
    // and of the prefetch one step old, every TD frames.  Draining ONCE here leaves the back edge as the only path with loads pending,
    // and the wait inside the loop becomes a counted one (stores and the younger prefetch stay in flight).
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0) (gfx9 encoding: expcnt 7, lgkmcnt 15 = no wait)
#endif
    for (int v0 = 0; v0 < total; v0 += TD) {
#pragma unroll
        for (int d = 0; d < TD; ++d) {
            const int v = v0 + d;                         // v >= total (last group only): computed, not stored
            const YuvRaw<T, C420, PX> cur0 = nx[d][0], cur1 = nx[d][1];
            prefetch(v + TD, nx[d][0], nx[d][1]);
            v2f rgb[3 * PX];
            yuv_pair_rgb<T, C420, STDM, PX>(cur0, cur1, a, g, g_fy, g_gy, g_fx0, g_gx0, rgb);
            eotf_apply_pairs<3 * PX, KIND>(rgb, a.e);
            v2f lum[PX];
#pragma unroll
            for (int i = 0; i < PX; ++i) lum[i] = lum_pair(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], a.w[0], a.w[1], a.w[2]);
            v2f acc_s[PX], acc_t[PX];                     // sustained / transient channel of (test, reference): set by the first tap
            karg_p tp = (karg_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(YuvArgs, taps2);
            if constexpr (FL > 8) asm volatile("" : "+s"(tp));      // reloaded per frame, see temporal_vec_body
            yuv_window_dispatch<FL, PX>(v & (FL - 1), win, lum, tp, acc_s, acc_t);
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < PX; ++i)
                s_t[lane * (PX + 1) + i] = make_float4(acc_s[i].x, acc_s[i].y, acc_t[i].x, acc_t[i].y);
            wave_lds_order();
#ifdef YUV_ABLATE_MEM
            const bool live = v < -1000000;
#else
            const bool live = (v >= FL - 1) && (v < total);
#endif
            // history frames and the steps past the end store nothing: their descriptor covers 0 bytes (a scalar select; the per-lane
            // offsets stay as they are -- round 6: -4 vector selects per frame)
            const __amdgpu_buffer_rsrc_t o = level_rsrc(l0_frame(a.out, max(v - (FL - 1), 0)), live ? frame_bytes : 0u);
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const int qq = min(i * 64 + lane, YUV_QUADS * PX - 1);
                const float4 val = s_t[(qq / PX + 1) * (PX + 1) + (qq % PX)];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{val.x, val.y, val.z, val.w}), o, soff[i], 0, 2 /*nt*/);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Pixels per lane of the vector kernel, per window length (4 or 2; A/B: tools/experiments/r6/s11.sh)
#ifndef YUV_PX8
#define YUV_PX8 4
#endif
#ifndef YUV_PX16
#define YUV_PX16 4
#endif
constexpr int yuv_px(int FL) { return FL == 8 ? YUV_PX8 : YUV_PX16; }
// waves per SIMD the register allocation aims at
#ifndef YUV_WAVES8
#define YUV_WAVES8 (YUV_PX8 == 4 ? 3 : 5)
#endif
#ifndef YUV_WAVES16
#define YUV_WAVES16 (YUV_PX16 == 4 ? 2 : 4)
#endif
#ifndef YUV_WAVES8_WIDE
#define YUV_WAVES8_WIDE (YUV_PX8 == 4 ? 2 : 4)      // 16-bit 4:4:4 (8 raw dwords per frame pair more at PX = 4), PQ (the longest display model)
#endif
// KIND = display model (compile-time: the host picks the instantiation): one loop body per kernel.  With a switch over the six
// bodies inside one kernel, scalar values of the prologue stayed alive across all of them and spilled (8 SGPRs in the 16-slot
// 4:2:0 kernels).
// Waves per workgroup (A/B switch): temporal_vec_kernel gains 4 % from 4 waves per workgroup on adjacent pixel blocks (K1_WPB8); this kernel
// does not -- 37.5-38.1 against 37.7-38.2 us per 4K frame, 1080p 10.5-11.2 against 10.2-10.4 (tools/experiments/r6/s5.sh), and the
// 8-bit 4:2:0 sRGB instantiation then spills one register -- so it stays at one wave per workgroup.
#ifndef YUV_WPB8
#define YUV_WPB8 1
#endif
constexpr int yuv_wpb(int FL) { return FL == 8 ? YUV_WPB8 : 1; }
template <int FL, typename T, bool C420, int KIND, bool STDM>
__global__ __launch_bounds__(64 * yuv_wpb(FL), (FL == 8 ? (((sizeof(T) == 2 && !C420) || KIND == FVVDP_EOTF_PQ) ? YUV_WAVES8_WIDE : YUV_WAVES8) : YUV_WAVES16))
void temporal_yuv_vec_kernel(const YuvArgs a_byval) {
    const YuvArgs& a = *(const YuvArgs*)__builtin_amdgcn_kernarg_segment_ptr();     // see temporal_vec_kernel
    (void)a_byval;
    constexpr int WPB = yuv_wpb(FL);
    constexpr int PX = yuv_px(FL);
    __shared__ float4 s_t_all[WPB][64 * (PX + 1)];
    const int wave = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int block = (int)blockIdx.x * WPB + wave;           // this wave's run of 62 pixel groups
    if constexpr (WPB > 1) {
        if (block * (YUV_QUADS * PX) >= a.W * a.H) return;     // the last workgroup of a frame whose block count is not a multiple
    }
    temporal_yuv_vec_body<FL, T, C420, KIND, STDM, PX>(a, s_t_all[wave], (int)(threadIdx.x & 63), block);
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
    L0Addr out;               // level 0 (see TemporalArgs::out)
    int* oob;
    const float* taps;   // device [2][fl]                        (used when inline_tables == 0)
    const int* idx;      // device [fl-1+n_out]
    int inline_tables;   // 1: short tables travel in the kernel arguments (no upload, no stream sync)
    float taps_i[2 * 32];
    int idx_i[T_MAX_IDX];
};

template <int SRC, int P>
__global__ __launch_bounds__(256) void temporal_generic_kernel(const GenericArgs a) {
    __shared__ float lutw[SRC == SRC_U8 ? 768 : 1];
    if constexpr (SRC == SRC_U8) {
        build_lutw(lutw, a.e, a.C, a.w, threadIdx.x, 256);
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
        const size_t off = (size_t)(a.inline_tables ? a.idx_i[t] : a.idx[t]) * a.frame_stride + p;
        S[0].lum(off, lt, bad);
        S[1].lum(off, lr, bad);
        *reinterpret_cast<float2*>(l0_frame(a.out, t) + (size_t)p * 2) = make_float2(lt[0], lr[0]);
    } else {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = a.fl - 1; k >= 0; --k) {
            const int fi = a.fl - 1 + t - k;
            const size_t off = (size_t)(a.inline_tables ? a.idx_i[fi] : a.idx[fi]) * a.frame_stride + p;
            float lt[1], lr[1];
            S[0].lum(off, lt, bad);
            S[1].lum(off, lr, bad);
            const float f0 = a.inline_tables ? a.taps_i[k] : a.taps[k];
            const float f1 = a.inline_tables ? a.taps_i[a.fl + k] : a.taps[a.fl + k];
            acc[0] = fmaf(lt[0], f0, acc[0]);
            acc[1] = fmaf(lr[0], f0, acc[1]);
            acc[2] = fmaf(lt[0], f1, acc[2]);
            acc[3] = fmaf(lr[0], f1, acc[3]);
        }
        *reinterpret_cast<float4*>(l0_frame(a.out, t) + (size_t)p * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    if (bad && a.oob) atomicOr(a.oob, 1);
}

// ---- luminance frames (first pass of the two-pass path for 33..64 taps) ------------------------------------------------
// Sample types / display models for which the 64-slot ring is not instantiated (k1_ring64_ok) used to take
// temporal_generic_kernel, which evaluates the display model fl times per pixel and output frame (float gray behind sRGB at
// 144 fps: 1126 us per 4K frame).  Instead: every source frame of the window is converted ONCE to fp32 luminance
// (L_test, L_ref: the same Sampler code, so the same values), and the 64-slot ring runs on those frames as it does for a
// user source's luminance frames.  One thread per pixel and frame; memory-bound.
struct LumArgs {
    const void* src[2];
    size_t chan_stride, frame_stride;
    int C, HW;
    EotfDev e;
    float w[3];
    int n_frames;
    float* out;              // [2 streams][n_frames][HW]
    int* oob;
    int fr[T_MAX_IDX];       // source frame of every output frame of this pass
};

template <int SRC>
__global__ __launch_bounds__(256) void luminance_frames_kernel(const LumArgs a) {
    __shared__ float lutw[SRC == SRC_U8 ? 768 : 1];
    if constexpr (SRC == SRC_U8) {
        build_lutw(lutw, a.e, a.C, a.w, threadIdx.x, 256);
        __syncthreads();
    }
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (p >= a.HW) return;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Sampler<SRC, 1> S;
        S.base = a.src[s];
        S.chan_stride = a.chan_stride;
        S.C = a.C;
        S.lutw = lutw;
        S.lut16 = a.e.lut;
        S.w0 = a.C == 3 ? a.w[0] : 1.0f;
        S.w1 = a.w[1];
        S.w2 = a.w[2];
        S.e = a.e;
        float l[1];
        S.lum((size_t)a.fr[k] * a.frame_stride + p, l, bad);
        a.out[((size_t)s * a.n_frames + k) * a.HW + p] = l[0];
    }
    if (bad && a.oob) atomicOr(a.oob, 1);
}

// the same first pass for planar YUV sources (33..64 taps): unpack, chroma upsampling, colour matrix, display model and
// luminance of every source frame of the window, once (yuv_lum above); the 64-slot ring then runs on the luminance frames
struct YuvLumArgs {
    YuvArgs y;               // src, frame_stride, geometry, conversion constants, display model (taps / idx unused)
    int n_frames;
    float* out;              // [2 streams][n_frames][HW]
    int fr[T_MAX_IDX];
};

template <typename T>
__global__ __launch_bounds__(256) void yuv_luminance_frames_kernel(const YuvLumArgs a) {
    const int HW = a.y.W * a.y.H;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (p >= HW) return;
    bool bad = false;
    const size_t off = (size_t)a.fr[k] * a.y.frame_stride;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const T* f = reinterpret_cast<const T*>(a.y.src[s]) + off;
        a.out[((size_t)s * a.n_frames + k) * HW + p] = yuv_lum<T>(f, a.y, p, bad);
    }
    if (bad && a.y.oob) atomicOr(a.y.oob, 1);
}

// planar [n][P][HW] <-> interleaved: frame f of a pyramid level at l0_frame(il, f), [HW][P]
template <int P>
__global__ void interleave_kernel(float* __restrict__ planar, const L0Addr il, int HW, int to_interleaved) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= HW) return;
    float* q = l0_frame(il, f) + (size_t)p * P;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        if (to_interleaved)
            q[k] = planar[((size_t)f * P + k) * HW + p];
        else
            planar[((size_t)f * P + k) * HW + p] = q[k];
    }
}

