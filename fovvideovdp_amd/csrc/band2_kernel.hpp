// Stage 2, two pyramid levels per pass.  Included by fvvdp_hip.hip after band_kernel.hpp.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// band2_kernel: reads Gaussian level A (= level i) once, keeps level B (= i+1) in registers, writes level C (= i+2),
// and accumulates sum(D^beta) of BOTH contrast bands i and i+1 in the same pass.
//
// Why: the one-level kernel (band_kernel) runs at the memory system's ceiling for a 4:1 read:write mix (any mix of
// reads and writes tops out near 4.9-5.0 TB/s on MI355X, pure reads reach 6.1-6.3: tools/microbench/mix.hip).  The only
// way to a shorter pass is fewer bytes: level B (1/4 of A) is neither written nor read back -- 4K video, levels
// 0+1: 207 MB -> 141 MB (+ strip halo) per frame.
//
// Work decomposition, as band_kernel: one single-wave workgroup streams down a strip; lane l owns level-B column
// J = 54*strip - 6 + l, i.e. level-A columns 2J, 2J+1, and (together with its pair lane l^1) level-C column K = J/2.
// Level A -> B is band_kernel's step (5 A rows in registers, DPP wave shifts for the horizontal taps).  Every second
// step ("stage k") one level-C row is produced from the last five level-B rows: vertical 5-tap thread-local,
// horizontal 5-tap over lanes l-2..l+2 (two DPP shifts), result broadcast inside the lane pair; then band B is
// evaluated for the two level-B rows 2k-2, 2k-1 (vertical expand from C rows k-2..k, horizontal from lanes l-2, l,
// l+2).  Lanes 6..59 own results (54 of 64: the two-level halo), so strips overlap by 20 level-A columns; rows: a
// chunk owns level-C rows [ka, kb) and runs 2 extra stages before and 1 after (7 level-A steps of halo).
//
// Borders follow the reference exactly as in band_kernel: symmetric rows for the reduce incl. the right-edge fix-up
// that gausspyr_reduce selects by the ROW-count parity (fvvdp_lpyr_dec.py:198-205), index clamping for the expand
// (fvvdp_lpyr_dec.py:126-142).  A level-B row past the bottom is the mirrored row (B[hb] = B[hb-1], B[hb+1] = B[hb-2]),
// which is what both the reduce (symmetric) and the expand (clamp, only B[hb]) ask for.
// ------------------------------------------------------------------------------------------------------------
#ifndef F2_PITCH
#define F2_PITCH 54         // level-B columns owned per wave: lanes 6..59, all the two-level halo leaves valid (the kernel is
                            // co-bound -- its arithmetic alone takes as long as its data flow, profiles/r03_pyramid_bounds.md: 54 vs a line-aligned 52 columns measured +3.4 %)
#define F2_HL 6             // halo lanes on the left (4 on the right); even, so lane parity == column parity
#endif

struct Band2Args {
    L0Addr A;               // level A, frame f at l0_frame(A, f): [h][w][P] (level 0 may live in two ranges, device_common.hpp)
    float* Gc;              // level C [n][hc][wc][P]
    int w, h, wb, hb, wc, hc;
    int n_strips, n_chunks, kr;       // kr = level-C rows per chunk
    int n_big, kr2, n_frames;         // chunks [0, n_big) have kr rows, chunks [n_big, n_chunks) kr2 rows (dispatched last, all frames)
    float mulA, mulB;       // band multipliers (lpyr.get_band, fvvdp_lpyr_dec.py:57-63)
    const float4* csfA;     // [32] slope-form 1-D CSF records of band A / band B (see BandArgs::csf)
    const float4* csfB;
    float y_first, y_inv_step, ly_lo, ly_hi;
    float lg_gain, lg_k, p, q0, q1, beta, lbkg_min, cmax, lg_dmax;
    float* partialA;        // [n][n_strips*n_chunks][2]
    float* partialB;
    int* tickets;           // nullptr: workgroup b takes item b (static split).  Else 16 zeroed counters -- [phase][XCD] -- and a grid larger
    int n_items;            //   than the n_items work items: see band2_kernel
};

// (dpp_reduce_taps / dpp_expand_taps: band_kernel.hpp)
// acc += u3*right(v); acc += u0*left2(v); acc += u1*left(v); acc += u4*right2(v); result of the even lane of each pair
__device__ __forceinline__ v2f dpp_reduce5_pair(v2f acc, v2f v, float u0, float u1, float u3, float u4) {
    float x = acc.x, y = acc.y, l1x, l1y, r1x, r1y, ox, oy;
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %2, %8" DPP_SHR "\n\t"
                 "v_mov_b32_dpp %3, %9" DPP_SHR "\n\t"
                 "v_mov_b32_dpp %4, %8" DPP_SHL "\n\t"
                 "v_mov_b32_dpp %5, %9" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %0, %8, %12" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %1, %9, %12" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %0, %2, %10" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %1, %3, %10" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %0, %8, %11" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %1, %9, %11" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %0, %4, %13" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %1, %5, %13" DPP_SHL "\n\t"
                 "s_nop 1\n\t"
                 "v_mov_b32_dpp %6, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_mov_b32_dpp %7, %1 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+v"(x), "+v"(y), "=&v"(l1x), "=&v"(l1y), "=&v"(r1x), "=&v"(r1y), "=&v"(ox), "=&v"(oy)
                 : "v"(v.x), "v"(v.y), "v"(u0), "v"(u1), "v"(u3), "v"(u4));
    return v2f{ox, oy};
}
// t += fl*left2(e); t += fr*right2(e)                                     (expand from lanes l-2, l+2)
__device__ __forceinline__ v2f dpp_expand2_taps(v2f t, v2f e, float fl, float fr) {
    float x = t.x, y = t.y, l1x, l1y, r1x, r1y;
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %2, %6" DPP_SHR "\n\t"
                 "v_mov_b32_dpp %3, %7" DPP_SHR "\n\t"
                 "v_mov_b32_dpp %4, %6" DPP_SHL "\n\t"
                 "v_mov_b32_dpp %5, %7" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %0, %2, %8" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %1, %3, %8" DPP_SHR "\n\t"
                 "v_fmac_f32_dpp %0, %4, %9" DPP_SHL "\n\t"
                 "v_fmac_f32_dpp %1, %5, %9" DPP_SHL
                 : "+v"(x), "+v"(y), "=&v"(l1x), "=&v"(l1y), "=&v"(r1x), "=&v"(r1y)
                 : "v"(e.x), "v"(e.y), "v"(fl), "v"(fr));
    return v2f{x, y};
}

#ifndef BAND2_LB
#define BAND2_LB 3
#endif
// INRANGE: the host has PROVEN from the display model, the RGB->Y weights and the temporal taps (luminance_range,
// fvvdp_hip.hip) that on these levels  (1) expand(ref sustained) >= lbkg_min, so L_bkg = max(., 0.1) is the identity
// (fvvdp_lpyr_dec.py:265),  (2) g - e < contrast_max * L_bkg for every plane, so the upper clamp of the contrast never binds
// (:266),  (3) L_bkg lies strictly inside the Y axis of the CSF table, so neither the clamp of the query (fvvdp.py:530) nor the
// clamp of the interval index binds.  The four clamps (7 VALU instructions of ~77 per band pixel, all of them min/max/med3 at
// 3.2-3.8 cycles on a saturated SIMD -- the kernel's arithmetic is as long as its data flow) are then dropped: same bits.
// Standard-dynamic-range displays qualify (standard_4k: luminances in [0.598, 200], clamp at >= 598 against a range of 253);
// HDR displays with a black level under 0.1 cd/m^2 and sources without a display model do not and take INRANGE = false.

template <int P, bool INRANGE>
__device__ __forceinline__ void band2_item(const Band2Args& a, const int strip, const int chunk, const int frame, const int lane,
                                           const float4 (*s_csf)[FVVDP_LUT_N], const bool in_step = false) {
    constexpr int HP = P / 2;
    const int blk = chunk * a.n_strips + strip;

    const int w = a.w, h = a.h, wb = a.wb, hb = a.hb, wc = a.wc, hc = a.hc;
    const int J = strip * F2_PITCH - F2_HL + lane;          // level-B column (may be < 0 or >= wb in halo lanes)
    const int K = J >> 1;                                    // level-C column of the lane pair
    const bool jeven = (lane & 1) == 0;
    const bool big = chunk < a.n_big;
    const int ka = big ? chunk * a.kr : a.n_big * a.kr + (chunk - a.n_big) * a.kr2;
    const int kb = min(ka + (big ? a.kr : a.kr2), hc);       // owned level-C rows
    const int ca = 2 * ka, cb = min(2 * kb, hb);             // owned level-B rows = level-A row pairs
    const bool owned = lane >= F2_HL && lane < F2_HL + F2_PITCH && J < wb;
    const int X0 = 2 * J, X1 = 2 * J + 1;
    const int xc0 = min(max(X0, 0), w - 1), xc1 = min(max(X1, 0), w - 1);
    const bool col1_ok = X1 < w;

    const float K0 = 0.05f, K1 = 0.25f, K2 = 0.4f, K3 = 0.25f, K4 = 0.05f;
    // ---- per-lane horizontal weights, level A -> B (column J) and B -> C (column K); see band_kernel ----------
    float wq0 = K0, wq1 = K1, wq2 = K2, wq3 = K3, wq4 = K4;
    if (J == 0) { wq2 += K1; wq3 += K0; wq0 = 0.0f; wq1 = 0.0f; }
    if (J == wb - 1) {
        const bool hodd = (h & 1) != 0;
        if (w & 1) { wq3 = 0.0f; wq4 = 0.0f; if (hodd) { wq2 += K3; wq1 += K4; } else { wq2 += K4; } }
        else { wq4 = 0.0f; if (hodd) { wq3 += K3; wq2 += K4; } else { wq3 += K4; } }
    }
    // taps of C column K at an even lane: B columns 2K-2 .. 2K+2 = lanes l-2, l-1, l, l+1, l+2
    float uq0 = K0, uq1 = K1, uq2 = K2, uq3 = K3, uq4 = K4;
    if (K == 0) { uq2 += K1; uq3 += K0; uq0 = 0.0f; uq1 = 0.0f; }
    if (K == wc - 1) {
        const bool hodd = (hb & 1) != 0;
        if (wb & 1) { uq3 = 0.0f; uq4 = 0.0f; if (hodd) { uq2 += K3; uq1 += K4; } else { uq2 += K4; } }
        else { uq4 = 0.0f; if (hodd) { uq3 += K3; uq2 += K4; } else { uq3 += K4; } }
    }
    // expand B -> A at columns X0 (even) / X1 (odd) from B columns J-1, J, J+1
    const bool at_l = (J <= 0), at_r = (J >= wb - 1);
    const float el = at_l ? 0.0f : 0.1f, er = at_r ? 0.0f : 0.1f;
    const float ec = 0.8f + (at_l ? 0.1f : 0.0f) + (at_r ? 0.1f : 0.0f);
    const float orr = at_r ? 0.0f : 0.5f, oc = at_r ? 1.0f : 0.5f;
    // expand C -> B at column J from C columns K-1, K, K+1 = lanes l-2, l, l+2 (every lane of a pair holds C[K])
    const bool k_l = (K <= 0), k_r = (K >= wc - 1);
    const float fl = (jeven && !k_l) ? 0.1f : 0.0f;
    const float fr = k_r ? 0.0f : (jeven ? 0.1f : 0.5f);
    const float fc = jeven ? (0.8f + (k_l ? 0.1f : 0.0f) + (k_r ? 0.1f : 0.0f)) : (k_r ? 1.0f : 0.5f);

    const float* Ga = l0_frame(a.A, frame);
    float* Gc = a.Gc + (size_t)frame * hc * wc * P;
    const __amdgpu_buffer_rsrc_t Gc_rsrc = level_rsrc(Gc, (unsigned int)(hc * wc * P) * 4u);

    // Level-A rows are read through a buffer resource covering the frame: the row's byte offset is wave-uniform (a scalar
    // register, computed by the scalar unit), the lane's column offset a loop-invariant vector register -- no vector
    // instruction goes into addressing (flat loads cost one 64-bit v_lshl_add_u64 each: 16 per loop iteration of a kernel
    // whose arithmetic is as long as its data flow).
#if defined(BAND2_ABLATE_MEM)      // profiling ablation: every wave re-reads 8 rows of frame 0 (L2 hits), nothing is stored
    const __amdgpu_buffer_rsrc_t Ga_rsrc = level_rsrc(a.A.lo, (unsigned int)(h * w * P) * 4u);
#else
    const __amdgpu_buffer_rsrc_t Ga_rsrc = level_rsrc(const_cast<float*>(Ga), (unsigned int)(h * w * P) * 4u);   // <= 133 MB per frame
#endif
    const unsigned int col0_b = (unsigned int)xc0 * (P * 4u), col1_b = (unsigned int)xc1 * (P * 4u);
    const unsigned int row_b = (unsigned int)w * (P * 4u);
    auto load_row = [&](int r, Px<P>& p0, Px<P>& p1) {
        int rr = r < 0 ? -1 - r : (r >= h ? 2 * h - 1 - r : r);   // symmetric padding (fvvdp_lpyr_dec.py:190-195)
        rr = min(max(rr, 0), h - 1);
#if defined(BAND2_ABLATE_MEM)
        rr &= 7;
#endif
        const unsigned int so = (unsigned int)rr * row_b;
        p0 = ld_px_buf<P>(Ga_rsrc, col0_b, so);
        p1 = ld_px_buf<P>(Ga_rsrc, col1_b, so);
    };


    // Level-A rows live in an 8-slot register ring: row r of the chunk sits in slot r & 7, a step's window is slots
    // base .. base+4 and the two rows of the next step are fetched into slots base+5, base+6.  The base advances by 2 per
    // step, so two stages (4 steps) bring it back: the stage body is instantiated for both phases and nothing is moved
    // (a 5-row window that is shifted costs 24 v_mov_b64 per stage; the kernel's arithmetic is as long as its data flow).
    Px<P> S[8][2];
    auto coarse_step = [&](auto base) -> Px<P> {          // level-B row from the 5-row window of level A at slot `base`
        constexpr int B = decltype(base)::value;
        Px<P>(&W0)[2] = S[(B + 0) & 7]; Px<P>(&W1)[2] = S[(B + 1) & 7]; Px<P>(&W2)[2] = S[(B + 2) & 7];
        Px<P>(&W3)[2] = S[(B + 3) & 7]; Px<P>(&W4)[2] = S[(B + 4) & 7];
        Px<P> c, va, vb;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f a0 = W0[0].h[k] * K0;
            a0 = pfma(W1[0].h[k], K1, a0);
            a0 = pfma(W2[0].h[k], K2, a0);
            a0 = pfma(W3[0].h[k], K3, a0);
            va.h[k] = pfma(W4[0].h[k], K4, a0);
            v2f b0 = W0[1].h[k] * K0;
            b0 = pfma(W1[1].h[k], K1, b0);
            b0 = pfma(W2[1].h[k], K2, b0);
            b0 = pfma(W3[1].h[k], K3, b0);
            vb.h[k] = pfma(W4[1].h[k], K4, b0);
        }
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f acc = va.h[k] * wq2;
            acc = pfma(vb.h[k], wq3, acc);
            c.h[k] = dpp_reduce_taps(acc, va.h[k], vb.h[k], wq0, wq1, wq4);
        }
        return c;
    };
    float accA[2] = {0.0f, 0.0f}, accB[2] = {0.0f, 0.0f};
    const float lg_base = a.lg_gain;
    const float lg_mask = a.lg_gain + a.lg_k;
    const float lg_bm[2] = {__log2f(a.mulA), __log2f(a.mulB)};
    const float pb = a.p * a.beta, pb_base = lg_base * pb, b_dmax = a.beta * a.lg_dmax, y_off = -a.y_first * a.y_inv_step;

    // per-pixel tail (contrast, CSF, masking, pooling term), see band_kernel::band_px for the derivation
    auto tail = [&](const Px<P>& g, const Px<P>& e, bool valid, const int band, float (&acc)[2]) {
#if defined(BAND2_ABLATE) && BAND2_ABLATE >= 1      // profiling ablation: no per-pixel tail, keep the data flow alive
        acc[0] += g.h[0].x + e.h[0].y + (valid ? 1.0f : 0.0f);
        (void)band;
        return;
#endif
        const float lb = INRANGE ? e.h[0].y : fmaxf(e.h[0].y, a.lbkg_min);
        const float dcap = a.cmax * lb;
        v2f d[HP];
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            // component-wise: the expanded level leaves the DPP blocks in single registers (no pair to subtract from)
            if constexpr (INRANGE) d[k] = v2f{g.h[k].x - e.h[k].x, g.h[k].y - e.h[k].y};
            else d[k] = v2f{fminf(g.h[k].x - e.h[k].x, dcap), fminf(g.h[k].y - e.h[k].y, dcap)};   // upper clamp only (fvvdp_lpyr_dec.py:266)
        }
        const float llb = fast_log2(lb);
        const float yq = INRANGE ? llb : __builtin_amdgcn_fmed3f(llb, a.ly_lo, a.ly_hi);
        const float t = fmaf(yq, a.y_inv_step, y_off);                       // (yq - y_first) * y_inv_step
        float4 r;
        float f;
        if constexpr (INRANGE) {
            r = s_csf[band][floor_to_int(t)];          // v_cvt_flr_i32_f32; t lies strictly inside (0, 31)
            f = __builtin_amdgcn_fractf(t);            // t - floor(t), one instruction
        } else {
            const float fi = __builtin_amdgcn_fmed3f(floorf(t), 0.0f, (float)(FVVDP_LUT_N - 2));
            r = s_csf[band][(int)fi];
            f = t - fi;
        }
        const float slog0 = fmaf(f, r.z, r.x), slog1 = fmaf(f, r.w, r.y);
        const float vm = valid ? 1.0f : 0.0f;
        const float lcn = lg_bm[band] - llb;
        if constexpr (HP == 2) {
            // beta * log2 D = beta*p*(ldiff + S') - beta*log2(1 + 2^(q*(lmin + S''))), S' = slog + lcn + lg_base, S'' = slog +
            // lcn + lg_mask: the constant factors are folded into fused multiply-adds (2 packed operations fewer per pixel
            // than the literal form; it differs from it by the rounding of the products only)
            const v2f sl = v2f{slog0, slog1};
            const v2f A = pfma(sl, pb, splat(fmaf(lcn, pb, pb_base)));        // beta*p*(slog + lcn + lg_base)
            const v2f lsm = sl + splat(lcn + lg_mask);
            const v2f ldiff = v2f{fast_log2(fabsf(d[0].x - d[0].y)), fast_log2(fabsf(d[1].x - d[1].y))};
            const v2f lmin = v2f{fast_log2(fminf(fabsf(d[0].x), fabsf(d[0].y))), fast_log2(fminf(fabsf(d[1].x), fabsf(d[1].y)))};
            const v2f ldb = pfma(ldiff, pb, A);                                // beta * p * (ldiff + S')
            const v2f lm = (lmin + lsm) * v2f{a.q0, a.q1};
            const v2f one_mq = v2f{fast_exp2(lm.x), fast_exp2(lm.y)} + splat(1.0f);
            const v2f tb = pfma(v2f{fast_log2(one_mq.x), fast_log2(one_mq.y)}, -a.beta, ldb);
            const v2f bl = v2f{fminf(tb.x, b_dmax), fminf(tb.y, b_dmax)};      // D <= d_max
            const v2f term = v2f{fast_exp2(bl.x), fast_exp2(bl.y)};
            const v2f av = __builtin_elementwise_fma(term, splat(vm), v2f{acc[0], acc[1]});
            acc[0] = av.x;
            acc[1] = av.y;
        } else {
            const float dT = d[0].x, dR = d[0].y;
            const float ls = slog0 + lcn;
            const float ld = a.p * (fast_log2(fabsf(dT - dR)) + (ls + lg_base));
            const float mq = fast_exp2(a.q0 * (fast_log2(fminf(fabsf(dT), fabsf(dR))) + (ls + lg_mask)));
            const float ldd = fminf(ld - fast_log2(1.0f + mq), a.lg_dmax);
            acc[0] = fmaf(fast_exp2(a.beta * ldd), vm, acc[0]);
        }
    };

    // band A for level-A rows 2c, 2c+1 (window rows 0, 1), expand from level-B rows c-1, c, c+1
    auto band_a_rows = [&](auto base, int c, const Px<P>& Bm1, const Px<P>& B0, const Px<P>& Bp1) {
        constexpr int B = decltype(base)::value;
        Px<P>(&W0)[2] = S[(B + 0) & 7]; Px<P>(&W1)[2] = S[(B + 1) & 7];
        Px<P> x00, x01, x10, x11, evE, evO;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f t = Bm1.h[k] * 0.1f;
            t = pfma(B0.h[k], 0.8f, t);
            evE.h[k] = pfma(Bp1.h[k], 0.1f, t);
            evO.h[k] = pfma(Bp1.h[k], 0.5f, B0.h[k] * 0.5f);
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
        tail(W0[0], x00, owned, 0, accA);
        tail(W0[1], x01, owned && col1_ok, 0, accA);
        tail(W1[0], x10, owned && row1_ok, 0, accA);
        tail(W1[1], x11, owned && row1_ok && col1_ok, 0, accA);
    };

    // ---- prologue: level-B row 2*ks and the window of the first step ---------------------------------------
    const int ks = max(ka - 2, 0);
    const bool last_chunk = kb >= hc;
    const int kend = last_chunk ? hc - 1 : kb;
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 4> I4;
    typedef std::integral_constant<int, 6> I6;
    {   // level-A rows 4ks-2 .. 4ks+2 into slots 6, 7, 0, 1, 2: row 4ks + j sits in slot j
        const int r0 = 4 * ks - 2;
#pragma unroll
        for (int k = 0; k < 5; ++k) load_row(r0 + k, S[(6 + k) & 7][0], S[(6 + k) & 7][1]);
    }
    Px<P> R[5];                       // level-B rows 2k-2 .. 2k+2 of the running stage (newest last)
    R[4] = coarse_step(I6());
    R[0] = R[1] = R[2] = R[3] = R[4];
    Px<P> CH[3];                      // level-C rows k-2, k-1, k
    CH[0] = CH[1] = CH[2] = R[4];
    load_row(4 * ks + 3, S[3][0], S[3][1]);
    load_row(4 * ks + 4, S[4][0], S[4][1]);

    // band B for level-B rows g0 (even), g1 (odd) from the level-C history
    auto band_b_rows = [&](const Px<P>& g0, const Px<P>& g1, int row0) {
        Px<P> evE, evO, e0, e1;
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            v2f t = CH[0].h[k] * 0.1f;
            t = pfma(CH[1].h[k], 0.8f, t);
            evE.h[k] = pfma(CH[2].h[k], 0.1f, t);
            evO.h[k] = pfma(CH[2].h[k], 0.5f, CH[1].h[k] * 0.5f);
        }
#pragma unroll
        for (int k = 0; k < HP; ++k) {
            e0.h[k] = dpp_expand2_taps(evE.h[k] * fc, evE.h[k], fl, fr);
            e1.h[k] = dpp_expand2_taps(evO.h[k] * fc, evO.h[k], fl, fr);
        }
        tail(g0, e0, owned && row0 < hb, 1, accB);
        tail(g1, e1, owned && (row0 + 1) < hb, 1, accB);
    };

    // ---- main loop: stage k = steps c = 2k, 2k+1, then level-C row k and band B of level-B rows 2k-2, 2k-1 ------
    auto stage = [&](auto basea, auto baseb, int k) {     // basea / baseb: ring slot of level-A row 4k / 4k+2
        constexpr int BA = decltype(basea)::value, BB = decltype(baseb)::value;
        Px<P> Be, Bo;
        // the waves of a workgroup walk ADJACENT strips of the same chunk: kept in step, their row requests of a stage cover one
        // contiguous piece of 4 level-A rows and their halo columns meet in the CU's vector cache (profiles/r04_lockstep.md)
        // (a soft barrier across the workgroups of a whole row of strips -- a counter in L2, bounded spin -- costs 6 us per use and
        // slows the launch down at every interval tried: profiles/r04_lockstep.md, section 6)
        if (in_step) __builtin_amdgcn_s_barrier();
        {   // step c = 2k: level-B row 2k+1
            const int c = 2 * k;
            load_row(2 * c + 5, S[(BA + 5) & 7][0], S[(BA + 5) & 7][1]);
            load_row(2 * c + 6, S[(BA + 6) & 7][0], S[(BA + 6) & 7][1]);
            Be = coarse_step(basea);
            if (c + 1 >= hb) Be = R[4];                               // B[hb] = B[hb-1]
            if (c >= ca && c < cb) band_a_rows(basea, c, R[3], R[4], Be);
        }
        {   // step c = 2k+1: level-B row 2k+2
            const int c = 2 * k + 1;
            load_row(2 * c + 5, S[(BB + 5) & 7][0], S[(BB + 5) & 7][1]);
            load_row(2 * c + 6, S[(BB + 6) & 7][0], S[(BB + 6) & 7][1]);
            Bo = coarse_step(baseb);
            if (c + 1 >= hb) Bo = (c + 1 == hb) ? Be : R[3];          // B[hb] = B[hb-1];  B[hb+1] = B[hb-2]
            if (c >= ca && c < cb) band_a_rows(baseb, c, R[4], Be, Bo);
        }
        // level-B window of this stage: rows 2k-2 .. 2k+2 (top: rows -2, -1 mirror to 1, 0)
        R[0] = R[2];
        R[1] = R[3];
        R[2] = R[4];
        R[3] = Be;
        R[4] = Bo;
        if (k == 0) {
            R[0] = R[3];
            R[1] = R[2];
        }
        // level-C row k: vertical 5-tap on the lane's own level-B column, horizontal over lanes l-2 .. l+2
        Px<P> Cn;
#pragma unroll
        for (int q = 0; q < HP; ++q) {
            v2f v = R[0].h[q] * K0;
            v = pfma(R[1].h[q], K1, v);
            v = pfma(R[2].h[q], K2, v);
            v = pfma(R[3].h[q], K3, v);
            v = pfma(R[4].h[q], K4, v);
            Cn.h[q] = dpp_reduce5_pair(v * uq2, v, uq0, uq1, uq3, uq4);
        }
#if defined(BAND2_ABLATE_MEM)
        st_px(Gc_rsrc, (k < -1000000) ? 0u : FVVDP_NO_STORE, Cn);
#else
        st_px(Gc_rsrc, (k >= ka && k < kb && owned && jeven && K < wc) ? (unsigned int)(k * wc + K) * (P * 4u) : FVVDP_NO_STORE, Cn);
#endif
        CH[0] = CH[1];
        CH[1] = CH[2];
        CH[2] = Cn;
        if (k == 0) CH[1] = Cn;                                       // C[-1] = C[0] (index clamp of the expand)
        const int row0 = 2 * k - 2;
        if (row0 >= ca && row0 < cb) band_b_rows(R[0], R[1], row0);
    };
    {
        int k = ks;
        for (; k + 1 <= kend; k += 2) {
            stage(I0(), I2(), k);
            stage(I4(), I6(), k + 1);
        }
        if (k <= kend) stage(I0(), I2(), k);
    }
    // ---- bottom of the image: the last level-B rows against C[hc] = C[hc-1] ---------------------------------
    if (last_chunk) {
        const int row0 = 2 * hc - 2;
        CH[0] = CH[1];
        CH[1] = CH[2];
        if (row0 >= ca && row0 < cb) band_b_rows(R[2], R[3], row0);
    }

    const float a0 = wave_sum(accA[0]), a1 = wave_sum(accA[1]);
    const float b0 = wave_sum(accB[0]), b1 = wave_sum(accB[1]);
    if (lane == 0) {
        const size_t o = ((size_t)frame * (a.n_strips * a.n_chunks) + blk) * 2;
        a.partialA[o] = a0;
        a.partialA[o + 1] = a1;
        a.partialB[o] = b0;
        a.partialB[o + 1] = b1;
    }
}

// waves per workgroup of band2_kernel: 1 .. 4 (adjacent strips of one chunk and frame, one barrier per stage); the launch decides.
// 8 and 12 waves, or a second barrier per stage, were slower (profiles/r04_lockstep.md).
#define BAND2_WPB_MAX 4
#ifdef BAND2_TIMELINE      // profiling build: (start, end) of every workgroup on the 100 MHz wall clock + where it ran (tools/gpu_timeline.py)
__device__ unsigned long long g_band2_timeline[4 * 65536];
#endif

// Work distribution of band2_kernel: one workgroup (wpb adjacent strips x one chunk x one frame) per work item, in two phases of
// block indices -- the tall chunks of all frames, then the short chunks (the bottom rows of every frame, cut finer).  The launch
// ends when the last item does and the slots that finish earlier idle for up to one item (profiles/r04_lockstep.md: 12 % of the
// wave-time of a 4K x 60 launch); short items last make that wait short.  Within a phase XCD x (the hardware places workgroup b on
// XCD b mod 8) walks a contiguous range of (frame, chunk, strip group): neighbouring strips run on one XCD at about the same time
// and share its L2 (see band_kernel).
// Resident workgroups that take their items from per-XCD atomic queues and steal from the slowest XCD (the XCDs of one box
// finish their eighth 5-8 % apart) were built and measured 3.5 % SLOWER than the hardware's dispatch (the loop around the item
// costs registers: 13-18 scalar spills) -- profiles/r04_lockstep.md, section 4.
// With `tickets` (round 5) the SAME item order is handed out at run time: every XCD has its own counter per phase; a workgroup takes the
// next item of the XCD it runs on -- own tall chunks, then own short ones -- and when its XCD has none left the next item of another
// XCD (tall first).  The grid is ~12 % larger than the number of items and workgroups that find nothing exit at once, so an XCD that
// runs ahead executes more items than an eighth and one that lags fewer: the hardware's equal split of workgroups no longer fixes
// the split of work.  One atomic per workgroup at its START, no loop around the item (the kernel's registers are untouched); the
// partial sums are indexed by item, so the results do not depend on who ran what.
template <int P, bool INRANGE = false>
__global__ __launch_bounds__(64 * BAND2_WPB_MAX, BAND2_LB) void band2_kernel(const Band2Args a) {
    __shared__ float4 s_csf[2][FVVDP_LUT_N];
    __shared__ int s_item[2];
    const int lane = threadIdx.x & 63;
#ifdef BAND2_TIMELINE
    const unsigned long long tl_t0 = wall_clock64();
#endif
    const int wpb = (int)(blockDim.x >> 6);
    const int n_groups = a.n_strips / wpb;                    // (the launch makes wpb divide n_strips)
    const int first = n_groups * a.n_big * a.n_frames;        // workgroups of the first phase
    int bid;
    bool tall;
    if (a.tickets) {
        if (threadIdx.x == 0) {
#if defined(__gfx942__) || defined(__gfx950__)
            const int x = (int)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7;       // XCC_ID: the XCD this workgroup runs on (hwreg 20 exists on gfx942 / gfx950 only; 8 XCDs assumed, any value 0..7 is correct)
#else
            const int x = (int)(blockIdx.x & 7);                                        // other targets: the hardware's round-robin placement as the estimate
#endif
            const int nbp[2] = {first, a.n_items - first};
            int item = -1, ph_found = 0;
            for (int k = 0; k < 16 && item < 0; ++k) {
                // own tall, own short, the other XCDs' tall, the other XCDs' short
                const int ph = (k == 0) ? 0 : (k == 1 ? 1 : (k < 9 ? 0 : 1));
                const int y = (k < 2) ? x : ((x + (k < 9 ? k - 1 : k - 8)) & 7);
                const int q8 = nbp[ph] >> 3, r8 = nbp[ph] & 7, cnt = q8 + (y < r8 ? 1 : 0);
                int* tk = a.tickets + ph * 8 + y;
                if (__hip_atomic_load(tk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= cnt) continue;
                const int t = atomicAdd(tk, 1);
                if (t < cnt) { item = y * q8 + min(y, r8) + t; ph_found = ph; }
            }
            s_item[0] = item;
            s_item[1] = ph_found;
        }
        __syncthreads();
        bid = s_item[0];
        tall = s_item[1] == 0;
        if (bid < 0) return;                                  // (uniform: every thread of the workgroup reads the same value)
    } else {
        int v = (int)blockIdx.x, nb = first;
        tall = v < first;
        if (!tall) { v -= first; nb = (int)gridDim.x - first; }
        // XCD-aware work order within a phase
        const int q8 = nb >> 3, r8 = nb & 7, x = v & 7;
        bid = x * q8 + min(x, r8) + (v >> 3);
    }
    const int group = bid % n_groups;
    bid /= n_groups;
    const int per_frame = tall ? a.n_big : a.n_chunks - a.n_big;
    const int chunk = (tall ? 0 : a.n_big) + bid % per_frame;
    const int frame = bid / per_frame;
    const int strip = group * wpb + (int)(threadIdx.x >> 6);
    if (threadIdx.x < FVVDP_LUT_N) {
        s_csf[0][threadIdx.x] = a.csfA[threadIdx.x];
        s_csf[1][threadIdx.x] = a.csfB[threadIdx.x];
    }
    __syncthreads();
    band2_item<P, INRANGE>(a, strip, chunk, frame, lane, s_csf, wpb > 1);
#ifdef BAND2_TIMELINE
    if (lane == 0 && a.w >= 2560 && blockIdx.x * wpb + (threadIdx.x >> 6) < 65536) {       // (the large launch only)
        unsigned long long* t = g_band2_timeline + 4 * (size_t)(blockIdx.x * wpb + (threadIdx.x >> 6));
        t[0] = tl_t0;
        t[1] = wall_clock64();
        t[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) << 32) | (unsigned int)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);   // XCC_ID, HW_ID
        t[3] = ((unsigned long long)frame << 32) | (unsigned int)(chunk * a.n_strips + strip);
    }
#endif
}
