// Stage 2 kernel: one fused pass per pyramid level (reduce, expand, contrast, CSF, masking, pooling).
// Included by fvvdp_hip.hip (one translation unit).
#pragma once
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
#ifndef STRIP_J             // (overridable for one timing-only experiment, profiles/r03_fov_variants.md)
#define STRIP_J 60          // coarse columns produced per wave (64 lanes - 2 halo lanes each side)
#endif


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
// Loads through a buffer resource: voff = the lane's byte offset (vector register), soff = a wave-uniform byte offset (scalar
// register) -- the address needs no vector arithmetic.
template <int P>
__device__ __forceinline__ Px<P> ld_px_buf(__amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff);
template <>
__device__ __forceinline__ Px<4> ld_px_buf<4>(__amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff) {
    const v4f t = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    Px<4> p;
    p.h[0] = v2f{t.x, t.y};
    p.h[1] = v2f{t.z, t.w};
    return p;
}
template <>
__device__ __forceinline__ Px<2> ld_px_buf<2>(__amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff) {
    const v2f t = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    Px<2> p;
    p.h[0] = t;
    return p;
}
// The coarse level is written once and only read by the next launch: non-temporal stores (measured +2-3 % on the
// read+write mix of this kernel, tools/microbench/membw.hip).  The store goes through a buffer resource that covers the
// frame's coarse level: a lane (or a whole wave) that must not write passes an out-of-range offset, which the hardware
// drops.  This keeps the store in straight-line code: behind a branch, the compiler has to wait for vmcnt(0) at the
// top of the next step (loads and stores share one in-order counter on gfx9), i.e. every step would wait for the
// previous step's store to be acknowledged before its own prefetched rows are usable.
// (level_rsrc / FVVDP_NO_STORE: device_common.hpp)
#ifndef BAND_STORE_AUX
#define BAND_STORE_AUX 2    // cache policy bits of the coarse-level stores: 2 = nt (streaming)
#endif
__device__ __forceinline__ void st_px(__amdgpu_buffer_rsrc_t r, unsigned int byte_off, const Px<4>& a) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{a.h[0].x, a.h[0].y, a.h[1].x, a.h[1].y}), r, byte_off, 0, BAND_STORE_AUX);
}
__device__ __forceinline__ void st_px(__amdgpu_buffer_rsrc_t r, unsigned int byte_off, const Px<2>& a) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, a.h[0]), r, byte_off, 0, BAND_STORE_AUX);
}

// ---- horizontal taps with the DPP shift folded into the multiply-add (v_fmac_f32_dpp) ------------------------------
// The compiler keeps `v_mov_b32_dpp` + `v_pk_fma_f32` for update_dpp() followed by fmaf (its DPP combiner does not fold
// wave shifts on gfx950), i.e. one extra full-rate VALU instruction per neighbour value; the two-level kernel's arithmetic is as long as its data flow, so
// the taps are written out.  Hazard: a DPP source operand written by a VALU instruction needs 2 wait states, which the
// compiler cannot see inside an asm statement -> every block opens with `s_nop 1`, and temporaries produced inside a
// block are read at least 3 instructions later.  Accumulation order = the order of the fma chain it replaces.
#define DPP_SHR " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define DPP_SHL " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
// acc += w0*left(va); acc += w1*left(vb); acc += w4*right(va)            (reduce, taps 0, 1, 4)
__device__ __forceinline__ v2f dpp_reduce_taps(v2f acc, v2f va, v2f vb, float w0, float w1, float w4) {
    float x = acc.x, y = acc.y;
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %2, %6" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %1, %3, %6" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %0, %4, %7" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %1, %5, %7" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %0, %2, %8" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %1, %3, %8" DPP_SHL
                 : "+v"(x), "+v"(y)
                 : "v"(va.x), "v"(va.y), "v"(vb.x), "v"(vb.y), "v"(w0), "v"(w1), "v"(w4));
    return v2f{x, y};
}
// even += el*left(e); even += er*right(e); odd += orr*right(e)           (expand from columns J-1, J, J+1)
__device__ __forceinline__ void dpp_expand_taps(v2f e, float el, float er, float orr, v2f& even, v2f& odd) {
    float ex = even.x, ey = even.y, ox = odd.x, oy = odd.y;
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %4, %6" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %1, %5, %6" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %0, %4, %7" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %1, %5, %7" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %2, %4, %8" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %3, %5, %8" DPP_SHL
                 : "+v"(ex), "+v"(ey), "+v"(ox), "+v"(oy)
                 : "v"(e.x), "v"(e.y), "v"(el), "v"(er), "v"(orr));
    even = v2f{ex, ey};
    odd = v2f{ox, oy};
}

// Layout of a band's slice of the CSF table (foveated mode): [rho interval][ecc][Y] of float4 records, the ecc rows padded
// from 32 to FOV_ROW entries and every rho plane by 8 more.  ds_read_b128 serves a wave in four 16-lane groups and a group
// runs at full rate only if its lanes' addresses are equal or fall into different 16-byte slots of the 256-byte bank row
// (MI355X_MICROARCH.md, LDS): the slot of cell (r, e, y) is (8 r + 4 e + y) mod 16, so the cells a group typically touches --
// a few adjacent Y intervals in one or two adjacent ecc intervals -- never collide.  Unpadded (row = 32 entries = two full
// bank rows) every pair (e, y), (e + 1, y) did: one extra LDS cycle on almost every access (r2 PMC).
#ifndef FOV_ROW
#define FOV_ROW 36
#endif
#ifndef FOV_PLANE
#define FOV_PLANE (FVVDP_LUT_N * FOV_ROW + 8)
#endif

struct BandArgs {
    L0Addr F;               // fine level, frame f at l0_frame(F, f): [h][w][P] (level 0 may live in two ranges, device_common.hpp)
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
    const float4* sublut;   // per band: [rw rho intervals][32 ecc, rows of FOV_ROW][32 Y] (FOV_PLANE entries per interval) of {S_log0[i], S_log1[i], S_log0[i+1]-S_log0[i], S_log1[i+1]-S_log1[i]} (i = rho knot)
    const float* axes;      // [3][32] knots: Y_log, rho_log, ecc_sqrt
    int rw, i_lo;           // rho knots covered by the band's sub-LUT: [i_lo, i_lo+rw]
    const float* fix;       // device [n][2]: gaze in frame pixels, or gaze view direction in degrees (map mode)
    const float* mvx;       // map mode (user geometry): view direction x,y [h][w] in degrees and resolution
    const float* mvy;       //   magnification [h][w] of this band, evaluated by the caller with the user's
    const float* mrm;       //   geometry object; nullptr = stock geometry computed in-kernel
    const float4* rmap;     // stock geometry: per pixel PAIR (columns 2J, 2J+1) {fraction, interval} x 2 of the rho axis of
                            //   the band's LUT slice -- frame-invariant, built once per geometry by fov_rho_map_kernel
    int rmap_w;             //   pixel pairs per row of rmap = (w + 1) / 2
    float size_m0, size_m1, dist_m, cos_delta, delta_rad;
    float rho_band, rho_lo, rho_hi, ecc_lo, ecc_hi;
    float inv_step[3], first[3];   // uniform-grid estimates of the three axes
    float grid_off[3];             // -first * inv_step
    float frac_scale[3];           // step / (step + 1e-6): fraction inside an interval from the grid position (interp.py:16)
    int frame_w, frame_h;
};


// Foveated mode, frame-invariant part of the CSF query: the spatial frequency of a pixel is rho_band times the
// resolution magnification at its view angle (fvvdp.py:424-442, fvvdp_display_model.py:475-526) -- a function of the
// pixel position only.  Its place on the rho axis of the band's LUT slice (interval and fraction, interp.py:11-20)
// is evaluated ONCE per geometry here, with exactly the operations band_kernel used per pixel and frame before
// (4 transcendentals and ~25 VALU instructions per pixel and frame saved for 8 B/pixel of reads that stay in L2 with the frame-fastest work order).
struct RhoMapArgs {
    float4* out;            // [h][(w+1)/2] {f(2J), k(2J), f(2J+1), k(2J+1)}, k = (interval - i_lo) * FOV_PLANE * 16 as a float =
                            //   BYTE offset of the interval's (ecc, Y) plane of float4 entries in the band's LUT slice
    int w, h;
    float size_m0, size_m1, dist_m, cos_delta, delta_rad;
    float rho_band, rho_lo, rho_hi, first, inv_step;
    int i_lo, rw;
    const float* axis;      // [32] rho_log knots
    int plane_bytes;        // bytes of one rho plane of the slice this map indexes (FOV_PLANE * 16 for the one-level kernels)
    int base_bytes;         // byte offset of the band's first plane in the kernel's LDS (two-level kernel: band B follows band A)
    float2* out_px;         // != nullptr: one {f, k} record per PIXEL [h][w] instead of `out` (level B of the two-level kernel)
};
__global__ __launch_bounds__(256) void fov_rho_map_kernel(const RhoMapArgs a) {
    const int J = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int pw = (a.w + 1) / 2;
    if (J >= pw) return;
    const float kyb = a.size_m1 / (float)a.h / a.dist_m;
    const float yp = ((float)y + 0.5f) + (-(float)a.h / 2.0f);
    const float vy = atanf(-yp * kyb) * 57.29577951308232f;
    const float kx = a.size_m0 / (float)a.w / a.dist_m;
    float r[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int X = 2 * J + i;
        const float xa = ((float)X + 0.5f) + (-(float)a.w / 2.0f);
        const float vx = atanf(xa * kx) * 57.29577951308232f;
        const float va = fminf(__builtin_amdgcn_sqrtf(vx * vx + vy * vy), 89.9f) * 0.017453292519943295f;
        const float rm = a.cos_delta * fast_rcp(__cosf(va) * __cosf(va + a.delta_rad));
        const float rho = a.rho_band * rm;
        const float rq = fast_log2(fminf(fmaxf(rho, a.rho_lo), a.rho_hi));
        const int k = min(max((int)floorf((rq - a.first) * a.inv_step), a.i_lo), a.i_lo + a.rw - 1);
        const float x0 = a.axis[k], x1 = a.axis[k + 1 < FVVDP_LUT_N ? k + 1 : k];
        const float f = fmaxf((rq - x0) * (1.0f / (x1 - x0 + 0.000001f)), 0.0f);
        r[2 * i] = f;
        r[2 * i + 1] = (float)((k - a.i_lo) * a.plane_bytes + a.base_bytes);
    }
    if (a.out_px) {
        a.out_px[(size_t)y * a.w + 2 * J] = make_float2(r[0], r[1]);
        if (2 * J + 1 < a.w) a.out_px[(size_t)y * a.w + 2 * J + 1] = make_float2(r[2], r[3]);
    } else {
        a.out[(size_t)y * pw + J] = make_float4(r[0], r[1], r[2], r[3]);
    }
}


// (int)floorf(x) as ONE instruction (the compiler emits v_floor_f32 + v_cvt_i32_f32)
__device__ __forceinline__ int floor_to_int(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

#ifndef FOV_WPB
#define FOV_WPB 4            // foveated mode: 4 independent waves per workgroup share the band's LUT slice in LDS
#endif
#ifndef FOV_MINW
#define FOV_MINW 2
#endif
#ifndef FOV_MINW_LEAN
#define FOV_MINW_LEAN 3     // the stock-geometry instantiation (FOVM == 1): 168 VGPRs, 3 waves per SIMD
#endif
#ifndef FOV_PHASE
#define FOV_PHASE 2          // pixels whose LDS reads are batched: pairs (160 VGPRs, 3 waves per SIMD); 4 = all of a step (181 VGPRs)
#endif
extern __shared__ __attribute__((aligned(16))) float4 s_lut_dyn[];

// One work item of a level: the wave streams down strip `strip` of frame `frame`, coarse rows of chunk `chunk`.  The tables
// (s_csf; foveated: s_ax, the LUT slice and the row table in dynamic LDS) are loaded by the calling kernel.
template <int P, bool DBG, int FOVM>
__device__ __forceinline__ void band_item(const BandArgs& a, const int strip, const int chunk, const int frame, const int lane,
                                          const float4* s_csf, const float2* s_ax) {
    constexpr bool FOV = FOVM != 0;
    constexpr bool LUT_LDS = FOVM == 1 || FOVM == 3;
    // FOVM 1: stock geometry with the frame-invariant rho map (the fast path: no code for the other cases in the loop);
    // FOVM 3: LUT slice in LDS, user geometry maps or no rho map;  FOVM 2: LUT slice in global memory (any case, DBG)
    // (round 5: a variant of FOVM 1 without the four clamps of the Y / eccentricity query where the host proves them unreachable --
    // 54 of 1452 VALU instructions per 16 pixels -- measured no faster, same box: profiles/r05_fov_budget.md; not kept)
    constexpr bool LEAN = FOVM == 1;
    constexpr int HP = P / 2;   // (test, ref) pairs = temporal channels
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

    const float* Gf = l0_frame(a.F, frame);
    float* Gc = a.Gc + (size_t)frame * hc * wc * P;
    const __amdgpu_buffer_rsrc_t Gc_rsrc = level_rsrc(Gc, (unsigned int)(hc * wc * P) * 4u);     // <= 33 MB per frame

    // rows of the fine level through a buffer resource: scalar row offset + loop-invariant lane offset, no vector address
    // arithmetic (see ld_px_buf)
#if defined(BAND_ABLATE_MEM)       // profiling ablation: every wave re-reads 8 rows of frame 0 (L2 hits), nothing is stored
    const __amdgpu_buffer_rsrc_t Gf_rsrc = level_rsrc(a.F.lo, (unsigned int)(h * w * P) * 4u);
#else
    const __amdgpu_buffer_rsrc_t Gf_rsrc = level_rsrc(const_cast<float*>(Gf), (unsigned int)(h * w * P) * 4u);    // <= 133 MB per frame
#endif
    const unsigned int col0_b = (unsigned int)xc0 * (P * 4u), col1_b = (unsigned int)xc1 * (P * 4u);
    const unsigned int row_b = (unsigned int)w * (P * 4u);
    auto load_row = [&](int r, Px<P>& p0, Px<P>& p1) {
        int rr = r < 0 ? -1 - r : (r >= h ? 2 * h - 1 - r : r);   // symmetric padding (fvvdp_lpyr_dec.py:190-195)
        rr = min(max(rr, 0), h - 1);
#if defined(BAND_ABLATE_MEM)
        rr &= 7;
#endif
        const unsigned int so = (unsigned int)rr * row_b;
        p0 = ld_px_buf<P>(Gf_rsrc, col0_b, so);
        p1 = ld_px_buf<P>(Gf_rsrc, col1_b, so);
    };

    // The 5-row window of the vertical filter plus the two rows in flight for the next step live in a ring of 8 row slots
    // that is never moved: fine row r sits in slot (r - (2*ca - 4)) mod 8, and the main loop below is unrolled over the 4
    // positions the window can take in the ring (a window that is shifted by two rows per step costs 24 64-bit register
    // moves per step and lane, 6-10 % of the loop's VALU time).
    Px<P> R[8][2];
    using std::integral_constant;

    // one coarse row from the window that starts in slot S0: vertical 5-tap in registers, horizontal 5-tap across lanes
    auto coarse_step = [&](auto S0) -> Px<P> {
        constexpr int s0 = decltype(S0)::value;
        Px<P> c, va, vb;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f a0 = R[s0][0].h[k] * K0;
            a0 = pfma(R[(s0 + 1) & 7][0].h[k], K1, a0);
            a0 = pfma(R[(s0 + 2) & 7][0].h[k], K2, a0);
            a0 = pfma(R[(s0 + 3) & 7][0].h[k], K3, a0);
            va.h[k] = pfma(R[(s0 + 4) & 7][0].h[k], K4, a0);
            v2f b0 = R[s0][1].h[k] * K0;
            b0 = pfma(R[(s0 + 1) & 7][1].h[k], K1, b0);
            b0 = pfma(R[(s0 + 2) & 7][1].h[k], K2, b0);
            b0 = pfma(R[(s0 + 3) & 7][1].h[k], K3, b0);
            vb.h[k] = pfma(R[(s0 + 4) & 7][1].h[k], K4, b0);
        }
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f acc = va.h[k] * wq2;
            acc = pfma(vb.h[k], wq3, acc);
            c.h[k] = dpp_reduce_taps(acc, va.h[k], vb.h[k], wq0, wq1, wq4);
        }
        return c;
    };

    // ---- prologue: coarse rows ca-1 and ca --------------------------------------------------------------
    {
        const int r0 = 2 * (ca - 1) - 2;                              // slot 0
#pragma unroll
        for (int k = 0; k < 5; ++k) load_row(r0 + k, R[k][0], R[k][1]);
    }
    const Px<P> cA = coarse_step(integral_constant<int, 0>());
    load_row(2 * ca + 1, R[5][0], R[5][1]);
    load_row(2 * ca + 2, R[6][0], R[6][1]);
    const Px<P> cB = coarse_step(integral_constant<int, 2>());       // rows 2ca-2 .. 2ca+2
    st_px(Gc_rsrc, active ? (unsigned int)(ca * wc + J) * (P * 4u) : FVVDP_NO_STORE, cB);
    Px<P> Gm1 = (ca > 0) ? cA : cB;
    Px<P> G0 = cB;
    load_row(2 * ca + 3, R[7][0], R[7][1]);
    load_row(2 * ca + 4, R[0][0], R[0][1]);

    float acc[2] = {0.0f, 0.0f};

    // foveated constants of this lane's two fine columns
    float vxa = 0.0f, vxb = 0.0f, gx = 0.0f, gy = 0.0f;
    if constexpr (FOV) {
        // pix2view_direction (fvvdp_display_model.py:498-510) on the band grid, pixel centres at +0.5
        const float xa = ((float)X0 + 0.5f) + (-(float)w / 2.0f);
        const float xb = ((float)X1 + 0.5f) + (-(float)w / 2.0f);
        const float kx = a.size_m0 / (float)w / a.dist_m;
        vxa = atanf(xa * kx) * 57.29577951308232f;
        vxb = atanf(xb * kx) * 57.29577951308232f;
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

    // eccentricity^2 = (vx - gx)^2 + (vy - gy)^2: the column term is the same for every row of the work item, the row term the
    // same for both columns of a row -- computed once each, one add per pixel is left (fvvdp.py:431-433)
    const float dxa = vxa - gx, dxb = vxb - gx;
    const float dxa2 = dxa * dxa, dxb2 = dxb * dxb;
    (void)dxa2; (void)dxb2;

    const float lg_bm = __log2f(a.band_mul);
    const float lg_base = a.lg_gain;              // log2(S) = interp + log2(gain)      (fvvdp.py:447)
    const float lg_mask = a.lg_gain + a.lg_k;      // log2(k*S)
    // constant factors of the log-domain tail, folded into fused multiply-adds (see band2_kernel.hpp)
    const float pb = a.p * a.beta, pb_base = lg_base * pb, b_dmax = a.beta * a.lg_dmax, inv_beta = 1.0f / a.beta;
    const float y_off = -a.y_first * a.y_inv_step;

    // per-pixel tail: contrast, CSF, masking, pooling  (fvvdp_lpyr_dec.py:259-269, fvvdp.py:395-467)
    auto band_px = [&](const Px<P>& g, const Px<P>& e, bool valid, int y, int x, float vx, float vy, float res_mag,
                       float pre_fR = 0.0f, float pre_kR = -1.0f) {
        (void)x; (void)res_mag; (void)pre_fR; (void)pre_kR;
        const float lb = fmaxf(e.h[0].y, a.lbkg_min);                  // plane 1 = reference (sustained)
        // contrast = min((g-e)/lb, cmax) * m.  Dividing by lb>0 commutes with |.|, min and the clamp, so the
        // division is carried as -log2(lb) in the log domain below: no reciprocal, no per-plane multiply.
        const float dcap = a.cmax * lb;                                // (g-e)/lb <= cmax  <=>  g-e <= cmax*lb
        v2f d[HP];
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            // component-wise: the expanded level leaves the DPP blocks in single registers (no pair to subtract from)
            d[k] = v2f{fminf(g.h[k].x - e.h[k].x, dcap), fminf(g.h[k].y - e.h[k].y, dcap)};   // upper clamp only (fvvdp_lpyr_dec.py:266)
        }
        const float llb = fast_log2(lb);
        const float yq = __builtin_amdgcn_fmed3f(llb, a.ly_lo, a.ly_hi);          // = log2(clamp(lb, Y[0], Y[-1]))  (fvvdp.py:530)
        float slog[2] = {0.0f, 0.0f};
        if constexpr (!FOV) {
            // 1-D table over log2(L_bkg) (uniform knots): interval from the grid, value = v[i] + f*(v[i+1]-v[i])
            const float t = fmaf(yq, a.y_inv_step, y_off);               // (yq - y_first) * y_inv_step
            const float fi = __builtin_amdgcn_fmed3f(floorf(t), 0.0f, (float)(FVVDP_LUT_N - 2));
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
            const float eq = __builtin_amdgcn_sqrtf(fminf(fmaxf(ecc, a.ecc_lo), a.ecc_hi));
            // interval on each (uniform) axis from the grid, fraction from the stored knots incl. interp.py:16's +1e-6
            auto axis = [&](int ax, float q, int lo, int hi, int& k, float& f) {
                k = min(max(floor_to_int((q - a.first[ax]) * a.inv_step[ax]), lo), hi);
                const float2 kn = s_ax[ax * FVVDP_LUT_N + k];
                f = fmaxf((q - kn.x) * kn.y, 0.0f);
            };
            int kY, kR, kE;
            float fY, fR, fE;
            axis(0, yq, 0, FVVDP_LUT_N - 2, kY, fY);
            int soR;                               // entry offset of the rho interval's (ecc, Y) plane
            if (pre_kR >= 0.0f) {                  // rho axis: frame-invariant, from the map (wave-uniform branch)
                soR = (int)pre_kR >> 4;
                fR = pre_fR;
            } else {
                const float rho = a.rho_band * res_mag;
                const float rq = fast_log2(fminf(fmaxf(rho, a.rho_lo), a.rho_hi));
                axis(1, rq, a.i_lo, a.i_lo + a.rw - 1, kR, fR);
                soR = (kR - a.i_lo) * FOV_PLANE;
            }
            (void)kR;
            axis(2, eq, 0, FVVDP_LUT_N - 2, kE, fE);
            // LUT slice [rho interval][ecc][Y]: the Y and ecc neighbours of a cell sit at compile-time distances
            const int so = soR + kE * FOV_ROW + kY;
            constexpr int sj = 1, sk = FOV_ROW;
            float4 v00, v10, v01, v11;                                                      // v[dj][dk]
            if constexpr (LUT_LDS) {
                v00 = s_lut_dyn[so]; v10 = s_lut_dyn[so + sj]; v01 = s_lut_dyn[so + sk]; v11 = s_lut_dyn[so + sk + sj];
            } else {
                const float4* sb = a.sublut + so;
                v00 = sb[0]; v10 = sb[sj]; v01 = sb[sk]; v11 = sb[sk + sj];
            }
            const float gY = 1.0f - fY, gE = 1.0f - fE;
            // interp3 (interp.py:53-57), same association: rho blend, then Y, then ecc.  The LUT slice stores the rho
            // blend in slope form {v[i], v[i+1] - v[i]}: v[i] + f * dv instead of v[i] * (1-f) + v[i+1] * f (1 ulp)
            slog[0] = (fmaf(v00.z, fR, v00.x) * gY + fmaf(v10.z, fR, v10.x) * fY) * gE +
                      (fmaf(v01.z, fR, v01.x) * gY + fmaf(v11.z, fR, v11.x) * fY) * fE;
            slog[1] = (fmaf(v00.w, fR, v00.y) * gY + fmaf(v10.w, fR, v10.y) * fY) * gE +
                      (fmaf(v01.w, fR, v01.y) * gY + fmaf(v11.w, fR, v11.y) * fY) * fE;
        }
        const float vm = valid ? 1.0f : 0.0f;
        const float lcn = lg_bm - llb;                                   // log2(m / lb)
        // D = |T'-R'|^p / (1 + (k*min(|T'|,|R'|))^q), T' = T*S   (fvvdp.py:585-595), in the log2 domain.
        // Video: the two temporal channels are carried as one (sustained, transient) pair through packed fp32 ops;
        // only the transcendentals are per component.
        float ldd_dbg[2] = {0.0f, 0.0f};
        if constexpr (HP == 2) {
            const v2f sl = v2f{slog[0], slog[1]};
            const v2f A = pfma(sl, pb, splat(fmaf(lcn, pb, pb_base)));   // beta * p * log2(S * m / lb)
            const v2f lsm = sl + splat(lcn + lg_mask);                   // log2(k * S * m / lb)
            const v2f ldiff = v2f{fast_log2(fabsf(d[0].x - d[0].y)), fast_log2(fabsf(d[1].x - d[1].y))};
            const v2f lmin = v2f{fast_log2(fminf(fabsf(d[0].x), fabsf(d[0].y))), fast_log2(fminf(fabsf(d[1].x), fabsf(d[1].y)))};
            const v2f ldb = pfma(ldiff, pb, A);                           // beta * p * (log2|T'-R'|)
            const v2f lm = (lmin + lsm) * v2f{a.q0, a.q1};
            const v2f one_mq = v2f{fast_exp2(lm.x), fast_exp2(lm.y)} + splat(1.0f);
            const v2f tb = pfma(v2f{fast_log2(one_mq.x), fast_log2(one_mq.y)}, -a.beta, ldb);   // beta * log2 D
            const v2f bl = v2f{fminf(tb.x, b_dmax), fminf(tb.y, b_dmax)};                      // D <= d_max
            const v2f term = v2f{fast_exp2(bl.x), fast_exp2(bl.y)};      // D^beta for the spatial pooling (fvvdp.py:467,607)
            const v2f av = __builtin_elementwise_fma(term, splat(vm), v2f{acc[0], acc[1]});
            acc[0] = av.x;
            acc[1] = av.y;
            ldd_dbg[0] = bl.x * inv_beta;
            ldd_dbg[1] = bl.y * inv_beta;
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

    // Foveated mode, plain evaluation (no maps written): the pixels of a step in two phases (FOV_PHASE at a time), so that
    // their LDS reads (the 4 corners of the LUT cell per pixel) are issued back to back and waited for once,
    // instead of dependent LDS round trips per pixel.  The same formula as band_px above; it differs from it by rounding only
    // (fractions from the grid position, blends in slope form: ~2e-6 of an interval, 1 ulp per blend).
    struct FovQ {
        float4 v00, v10, v01, v11;
        float fY, fE, fR, llb;
        v2f d[HP];
    };
    auto fov_a = [&](const Px<P>& g, const Px<P>& e, float dx2, float dy2, float pre_fR, float pre_kR) -> FovQ {
        FovQ q;
        const float lb = fmaxf(e.h[0].y, a.lbkg_min);
        const float dcap = a.cmax * lb;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            // component-wise on purpose: the expanded level comes out of the DPP blocks as single registers, a packed
            // subtraction would need them copied into a register pair first
            q.d[k] = v2f{fminf(g.h[k].x - e.h[k].x, dcap), fminf(g.h[k].y - e.h[k].y, dcap)};
        }
        q.llb = fast_log2(lb);
        const float yq = __builtin_amdgcn_fmed3f(q.llb, a.ly_lo, a.ly_hi);
#ifdef FOV_ABLATE_ECC     // timing experiment only: what the eccentricity arithmetic costs (results are wrong)
        const float eq = dx2 + dy2;
#else
        const float ecc = __builtin_amdgcn_sqrtf(dx2 + dy2);
        const float eq = __builtin_amdgcn_sqrtf(__builtin_amdgcn_fmed3f(ecc, a.ecc_lo, a.ecc_hi));
#endif
        // Y and ecc axes are uniform: interval = floor of the grid position t, fraction = (t - interval) * step/(step+1e-6)
        // (interp.py:11-20 computes (q - knot)/(knot' - knot + 1e-6) from the stored knots: equal to ~2e-6 of an interval,
        // which band_px, the map-writing path, still does).  No LDS look-up, no dependent round trip before the cell reads.
        const float tY = fmaf(yq, a.inv_step[0], a.grid_off[0]);
        const float iY = __builtin_amdgcn_fmed3f(floorf(tY), 0.0f, (float)(FVVDP_LUT_N - 2));
        q.fY = (tY - iY) * a.frac_scale[0];
#ifdef FOV_ABLATE_ECC
        const float iE = 3.0f;
        q.fE = eq;
#else
        const float tE = fmaf(eq, a.inv_step[2], a.grid_off[2]);
        const float iE = __builtin_amdgcn_fmed3f(floorf(tE), 0.0f, (float)(FVVDP_LUT_N - 2));
        q.fE = (tE - iE) * a.frac_scale[2];
#endif
        // byte offset of the cell (rho plane + ecc * 512 + Y * 16) in float: small integers are exact, one conversion
        const int bo = (int)fmaf(iE, (float)(FOV_ROW * 16), fmaf(iY, 16.0f, pre_kR));
        q.fR = pre_fR;
        const float4* cell = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_lut_dyn) + bo);
        constexpr int sj = 1, sk = FOV_ROW;
#ifdef FOV_ABLATE_LDS      // timing experiment only: no LUT reads (results are wrong)
        (void)cell; (void)sj; (void)sk;
        q.v00 = make_float4(q.fY, q.fE, 0.1f, 0.2f);
        q.v10 = make_float4(q.fE, q.fY, 0.2f, 0.1f);
        q.v01 = make_float4(q.fY, q.fY, 0.3f, 0.1f);
        q.v11 = make_float4(q.fE, q.fE, 0.1f, 0.3f);
#else
        q.v00 = cell[0];
        q.v10 = cell[sj];
        q.v01 = cell[sk];
        q.v11 = cell[sk + sj];
#endif
        return q;
    };
    auto fov_b = [&](const FovQ& q, bool valid) {
        const float fY = q.fY, fE = q.fE, fR = q.fR;
        // interp3 (interp.py:53-57), same association (rho, then Y, then ecc); both temporal channels as one packed pair;
        // every blend in slope form a + f * (b - a) (the reference's a * (1 - f) + b * f to 1 ulp)
        auto rho_blend = [&](const float4& v) { return pfma(v2f{v.z, v.w}, fR, v2f{v.x, v.y}); };
        const v2f r00 = rho_blend(q.v00), r10 = rho_blend(q.v10), r01 = rho_blend(q.v01), r11 = rho_blend(q.v11);
        const v2f y0 = pfma(r10 - r00, fY, r00), y1 = pfma(r11 - r01, fY, r01);
        const v2f sl2 = pfma(y1 - y0, fE, y0);
        const float s0 = sl2.x, s1 = sl2.y;
        const float vm = valid ? 1.0f : 0.0f;
        const float lcn = lg_bm - q.llb;
        if constexpr (HP == 2) {
            const v2f sl = v2f{s0, s1};
            const v2f A = pfma(sl, pb, splat(fmaf(lcn, pb, pb_base)));
            const v2f lsm = sl + splat(lcn + lg_mask);
            const v2f ldiff = v2f{fast_log2(fabsf(q.d[0].x - q.d[0].y)), fast_log2(fabsf(q.d[1].x - q.d[1].y))};
            const v2f lmin = v2f{fast_log2(fminf(fabsf(q.d[0].x), fabsf(q.d[0].y))), fast_log2(fminf(fabsf(q.d[1].x), fabsf(q.d[1].y)))};
            const v2f ldb = pfma(ldiff, pb, A);
            const v2f lm = (lmin + lsm) * v2f{a.q0, a.q1};
            const v2f one_mq = v2f{fast_exp2(lm.x), fast_exp2(lm.y)} + splat(1.0f);
            const v2f tb = pfma(v2f{fast_log2(one_mq.x), fast_log2(one_mq.y)}, -a.beta, ldb);
            const v2f bl = v2f{fminf(tb.x, b_dmax), fminf(tb.y, b_dmax)};
            const v2f term = v2f{fast_exp2(bl.x), fast_exp2(bl.y)};
            const v2f av = __builtin_elementwise_fma(term, splat(vm), v2f{acc[0], acc[1]});
            acc[0] = av.x;
            acc[1] = av.y;
        } else {
            const float dT = q.d[0].x, dR = q.d[0].y;
            const float ls = s0 + lcn;
            const float ld = a.p * (fast_log2(fabsf(dT - dR)) + (ls + lg_base));
            const float mq = fast_exp2(a.q0 * (fast_log2(fminf(fabsf(dT), fabsf(dR))) + (ls + lg_mask)));
            const float ldd = fminf(ld - fast_log2(1.0f + mq), a.lg_dmax);
            acc[0] = fmaf(fast_exp2(a.beta * ldd), vm, acc[0]);
        }
    };

    // ---- main loop: band rows 2c, 2c+1 for c in [ca, cb) ------------------------------------------------
    auto step = [&](auto PH, const int c) {
        // window = fine rows 2c .. 2c+4 in slots s0 .. s0+4 (mod 8)
        constexpr int s0 = (4 + 2 * decltype(PH)::value) & 7;
        const Px<P> (&W0)[2] = R[s0];
        const Px<P> (&W1)[2] = R[(s0 + 1) & 7];
        // prefetch the two rows of the next step.  Unconditional (the last step re-reads two rows it does not use; the row
        // index is clamped into the image): with the loads and the store in straight-line code the wait at the top of
        // the next step is vmcnt(1) -- rows landed, store still in flight -- instead of vmcnt(0).
        // foveated, stock geometry: the rho-map records of this step's two rows are requested BEFORE the row prefetch.
        // Loads return in order (vmcnt): behind the prefetch, the wait for the map was a wait for the two rows just requested
        // from HBM -- every step, with 3 waves per SIMD to cover it.
        float4 ra = make_float4(0.0f, -1.0f, 0.0f, -1.0f), rb = ra;      // {f, k} of columns X0, X1 in rows 2c, 2c+1
        if constexpr (FOV) {
            if (LEAN || (a.rmap && !a.mvx)) {
                const int jj = min(max(J, 0), a.rmap_w - 1);
                ra = a.rmap[(size_t)min(2 * c, h - 1) * a.rmap_w + jj];
                rb = a.rmap[(size_t)min(2 * c + 1, h - 1) * a.rmap_w + jj];
            }
        }
        float mvx4[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mvy4[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mrm4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if constexpr (FOV && !LEAN) {
            if (a.mvx) {                      // user geometry: maps evaluated by the caller, requested first for the same reason
                const int ya = min(2 * c, h - 1), yb = min(2 * c + 1, h - 1);
                const size_t o[4] = {(size_t)ya * w + xc0, (size_t)ya * w + xc1, (size_t)yb * w + xc0, (size_t)yb * w + xc1};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    mvx4[i] = a.mvx[o[i]];
                    mvy4[i] = a.mvy[o[i]];
                    mrm4[i] = a.mrm[o[i]];
                }
            }
        }
#if defined(FOV_PRIO)       // A/B build (profiles/r06_fov_floor.md): the memory-issuing head of a step at raised wave priority
        if constexpr (FOV) __builtin_amdgcn_s_setprio(FOV_PRIO);
#endif
        if constexpr (FOV) __builtin_amdgcn_sched_barrier(0);
        load_row(2 * c + 5, R[(s0 + 5) & 7][0], R[(s0 + 5) & 7][1]);
        load_row(2 * c + 6, R[(s0 + 6) & 7][0], R[(s0 + 6) & 7][1]);
        const Px<P> cN = coarse_step(integral_constant<int, s0>());       // coarse row c+1
        const bool has_next = (c + 1) <= (hc - 1);
        Px<P> Gp1 = has_next ? cN : G0;       // index clamp of the expand (fvvdp_lpyr_dec.py:134,138)
#if defined(BAND_ABLATE_MEM)
        st_px(Gc_rsrc, (c < -1000000) ? 0u : FVVDP_NO_STORE, cN);
#else
        st_px(Gc_rsrc, (has_next && (c + 1) < cb && active) ? (unsigned int)((c + 1) * wc + J) * (P * 4u) : FVVDP_NO_STORE, cN);
#endif
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
            x00.h[k] = evE.h[k] * ec;
            x01.h[k] = evE.h[k] * oc;
            dpp_expand_taps(evE.h[k], el, er, orr, x00.h[k], x01.h[k]);
            x10.h[k] = evO.h[k] * ec;
            x11.h[k] = evO.h[k] * oc;
            dpp_expand_taps(evO.h[k], el, er, orr, x10.h[k], x11.h[k]);
        }
        const bool row1_ok = (2 * c + 1) < h;
#if defined(FOV_PRIO)
#if !defined(FOV_TAILPRIO)
#define FOV_TAILPRIO 0      // (the reverse experiment: the arithmetic tail at raised priority)
#endif
        if constexpr (FOV) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(FOV_TAILPRIO); }
#endif
#if defined(BAND_ABLATE) && BAND_ABLATE >= 1      // profiling ablation: no per-pixel tail, keep the data flow alive
        acc[0] += x00.h[0].x + x01.h[0].x + x10.h[0].x + x11.h[0].x + W0[0].h[0].x + W0[1].h[0].x + W1[0].h[0].x + W1[1].h[0].x;
        if (false)
#endif
        {
        float vy0 = 0.0f, vy1 = 0.0f;        // vertical view angle of the two fine rows (foveated)
        if constexpr (FOV) {
            const float* s_vy = reinterpret_cast<const float*>(s_lut_dyn + (LUT_LDS ? FOV_PLANE * a.rw : 0));
            vy0 = s_vy[2 * c];
            vy1 = s_vy[min(2 * c + 1, h - 1)];
        }
        if constexpr (FOV) {
            const float dy0 = vy0 - gy, dy1 = vy1 - gy;
            [[maybe_unused]] const float dy02 = dy0 * dy0, dy12 = dy1 * dy1;      // row terms of the squared eccentricity (stock geometry)
            float vx4[4] = {vxa, vxb, vxa, vxb}, vy4[4] = {vy0, vy0, vy1, vy1}, rm4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (!LEAN && a.mvx) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vx4[i] = mvx4[i];
                    vy4[i] = mvy4[i];
                    rm4[i] = mrm4[i];
                }
            } else if (!LEAN && !a.rmap) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float va = fminf(__builtin_amdgcn_sqrtf(vx4[i] * vx4[i] + vy4[i] * vy4[i]), 89.9f) * 0.017453292519943295f;
                    rm4[i] = a.cos_delta * fast_rcp(__cosf(va) * __cosf(va + a.delta_rad));
                }
            }
            if constexpr (LUT_LDS && !DBG) {
                if (FOV_PHASE != 0 && (LEAN || (a.rmap && !a.mvx))) {   // stock geometry: phased evaluation (see fov_a / fov_b)
#if FOV_PHASE == 4
                    const FovQ q0 = fov_a(W0[0], x00, dxa2, dy02, ra.x, ra.y);
                    const FovQ q1 = fov_a(W0[1], x01, dxb2, dy02, ra.z, ra.w);
                    const FovQ q2 = fov_a(W1[0], x10, dxa2, dy12, rb.x, rb.y);
                    const FovQ q3 = fov_a(W1[1], x11, dxb2, dy12, rb.z, rb.w);
                    fov_b(q0, active);
                    fov_b(q1, active && col1_ok);
                    fov_b(q2, active && row1_ok);
                    fov_b(q3, active && row1_ok && col1_ok);
#else               // two pixels per phase (default): half the registers in flight; the scheduler may not interleave the phases
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const FovQ q0 = fov_a(W0[0], x00, dxa2, dy02, ra.x, ra.y);
                        const FovQ q1 = fov_a(W0[1], x01, dxb2, dy02, ra.z, ra.w);
                        fov_b(q0, active);
                        fov_b(q1, active && col1_ok);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const FovQ q2 = fov_a(W1[0], x10, dxa2, dy12, rb.x, rb.y);
                        const FovQ q3 = fov_a(W1[1], x11, dxb2, dy12, rb.z, rb.w);
                        fov_b(q2, active && row1_ok);
                        fov_b(q3, active && row1_ok && col1_ok);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#endif
                } else {
                    band_px(W0[0], x00, active, 2 * c, X0, vx4[0], vy4[0], rm4[0], ra.x, ra.y);
                    band_px(W0[1], x01, active && col1_ok, 2 * c, X1, vx4[1], vy4[1], rm4[1], ra.z, ra.w);
                    band_px(W1[0], x10, active && row1_ok, 2 * c + 1, X0, vx4[2], vy4[2], rm4[2], rb.x, rb.y);
                    band_px(W1[1], x11, active && row1_ok && col1_ok, 2 * c + 1, X1, vx4[3], vy4[3], rm4[3], rb.z, rb.w);
                }
            } else {
                band_px(W0[0], x00, active, 2 * c, X0, vx4[0], vy4[0], rm4[0], ra.x, ra.y);
                band_px(W0[1], x01, active && col1_ok, 2 * c, X1, vx4[1], vy4[1], rm4[1], ra.z, ra.w);
                band_px(W1[0], x10, active && row1_ok, 2 * c + 1, X0, vx4[2], vy4[2], rm4[2], rb.x, rb.y);
                band_px(W1[1], x11, active && row1_ok && col1_ok, 2 * c + 1, X1, vx4[3], vy4[3], rm4[3], rb.z, rb.w);
            }
        } else {
            band_px(W0[0], x00, active, 2 * c, X0, 0.0f, 0.0f, 1.0f);
            band_px(W0[1], x01, active && col1_ok, 2 * c, X1, 0.0f, 0.0f, 1.0f);
            band_px(W1[0], x10, active && row1_ok, 2 * c + 1, X0, 0.0f, 0.0f, 1.0f);
            band_px(W1[1], x11, active && row1_ok && col1_ok, 2 * c + 1, X1, 0.0f, 0.0f, 1.0f);
        }
        }
        Gm1 = G0;
        G0 = Gp1;
    };
    for (int c = ca; c < cb; c += 4) {        // wave-uniform trip count
        step(integral_constant<int, 0>(), c);
        if (c + 1 >= cb) break;
        step(integral_constant<int, 1>(), c + 1);
        if (c + 2 >= cb) break;
        step(integral_constant<int, 2>(), c + 2);
        if (c + 3 >= cb) break;
        step(integral_constant<int, 3>(), c + 3);
    }

    const float s0 = wave_sum(acc[0]);
    const float s1 = wave_sum(acc[1]);
    if (lane == 0) {
        float* o = a.partial + ((size_t)frame * (a.n_strips * a.n_chunks) + blk) * 2;
        o[0] = s0;
        o[1] = s1;
    }
}

// Tables of one level into LDS, by all `nthreads` threads of the workgroup (no barrier here): the 1-D CSF records, or in
// foveated mode the axis knots, the band's LUT slice and the vertical view angle of every band row.
template <int FOVM>
__device__ __forceinline__ void band_load_tables(const BandArgs& a, float4* s_csf, float2* s_ax, const int tid, const int nthreads) {
    constexpr bool FOV = FOVM != 0;
    constexpr bool LUT_LDS = FOVM == 1 || FOVM == 3;
    if constexpr (!FOV) {
        if (tid < FVVDP_LUT_N) s_csf[tid] = a.csf[tid];
    } else {
        for (int i = tid; i < 3 * FVVDP_LUT_N; i += nthreads) {
            const int k = i % FVVDP_LUT_N;
            const float x0 = a.axes[i];
            const float x1 = a.axes[k + 1 < FVVDP_LUT_N ? i + 1 : i];
            s_ax[i] = make_float2(x0, 1.0f / (x1 - x0 + 0.000001f));
        }
        if constexpr (LUT_LDS) {
            const int nl = FOV_PLANE * a.rw;
            for (int i = tid; i < nl; i += nthreads) s_lut_dyn[i] = a.sublut[i];
        }
        // vertical view angle of every band row (pix2view_direction, fvvdp_display_model.py:498-510): one atan per
        // row and workgroup instead of two per wave and step
        float* s_vy = reinterpret_cast<float*>(s_lut_dyn + (LUT_LDS ? FOV_PLANE * a.rw : 0));
        const float kyb = a.size_m1 / (float)a.h / a.dist_m;
        for (int i = tid; i < a.h; i += nthreads) {
            const float yp = ((float)i + 0.5f) + (-(float)a.h / 2.0f);
            s_vy[i] = atanf(-yp * kyb) * 57.29577951308232f;
        }
    }
}

// FOVM: 0 = non-foveated, 1 = foveated with the band's LUT slice in (dynamic) LDS, 2 = foveated, LUT slice in global
// memory (slice too large, or the map-writing variant).  A compile-time choice: with a run-time flag the compiler
// merges the two look-ups into one flat load, which is slower than ds_read_b128.
#ifndef FOV_FRAME_FASTEST
#define FOV_FRAME_FASTEST 1
#endif
template <int P, bool DBG, int FOVM>
__global__ __launch_bounds__(FOVM ? 64 * FOV_WPB : 64, FOVM ? (FOVM == 1 ? FOV_MINW_LEAN : FOV_MINW) : (DBG ? 2 : 4)) void band_kernel(const BandArgs a_byval) {
    // the argument block is read from the kernel-argument segment where it is needed (scalar loads) instead of being held in
    // scalar registers from the top of the kernel: the variants with many arguments in use (difference maps, caller-built view
    // maps) otherwise keep 13-48 of them in vector-register lanes
    const BandArgs& a = *(const BandArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_byval;
    constexpr bool FOV = FOVM != 0;
    constexpr int WPB = FOV ? FOV_WPB : 1;
    __shared__ float4 s_csf[FVVDP_LUT_N];
    __shared__ float2 s_ax[FOV ? 3 * FVVDP_LUT_N : 1];     // {knot k, 1/(knot k+1 - knot k + 1e-6)} of the three axes

    const int lane = FOV ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    // XCD-aware work order: hardware places workgroup b on XCD b % 8 (speed only, never correctness).  Give each
    // XCD a contiguous range of work items (strip fastest, then chunk, then frame) so that neighbouring strips,
    // which share their 4+4 halo columns, run on the same XCD at about the same time and hit in its L2.
    int bid;
    {
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
        bid = x * q8 + min(x, r8) + (blockIdx.x >> 3);
        // the wave number is the same in all lanes: say so, otherwise every quantity derived from the work item (rows,
        // loop counter, row addresses, the store descriptor) lives in vector registers and is recomputed by the VALU
        if constexpr (FOV) bid = bid * WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    }
    const bool wave_has_work = !FOV || bid < a.n_items;
    int strip, chunk, frame;
    if constexpr (FOVM == 1 && FOV_FRAME_FASTEST) {
        // frame fastest: an XCD walks all frames of a tile before the next tile, the tile's slice of the (frame-invariant)
        // rho map stays in that XCD's L2 instead of being fetched once per frame
        const int n_tiles = a.n_strips * a.n_chunks, n_frames = a.n_items / n_tiles;
        frame = bid % n_frames;
        bid /= n_frames;
        strip = bid % a.n_strips;
        chunk = bid / a.n_strips;
    } else {
        strip = bid % a.n_strips;
        bid /= a.n_strips;
        chunk = bid % a.n_chunks;
        frame = bid / a.n_chunks;
    }
    band_load_tables<FOVM>(a, s_csf, s_ax, (int)threadIdx.x, 64 * WPB);
    __syncthreads();
    if constexpr (FOV) {
        if (!wave_has_work) return;
    }
    band_item<P, DBG, FOVM>(a, strip, chunk, frame, lane, s_csf, s_ax);
}
