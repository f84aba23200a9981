// Small kernels: pooled-sum finalisation, heat-map reconstruction.  Included by fvvdp_hip.hip.
#pragma once
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

// one wave: Q of (band b, temporal channel cc, frame slot s); lane l adds partials l, l+64, ... in fp64, then a fixed shuffle tree
__device__ __forceinline__ void finalize_one(const FinalizeArgs& a, const int b, const int cc, const int s, const int lane) {
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

__global__ __launch_bounds__(64) void finalize_kernel(const FinalizeArgs a) {
    // one wave per (band, cc, slot)
    const int i = blockIdx.x;
    finalize_one(a, i / (2 * a.n), (i / a.n) % 2, i % a.n, (int)threadIdx.x);
}

// Pooling of the per-band, per-channel, per-frame differences into one JOD value (do_pooling_and_jods,
// fvvdp.py:337-357): Minkowski sums over bands (beta_sch), temporal channels (beta_tch, transient weighted by
// w_transient) and frames (beta_t, normalised), then the JOD regression.  One workgroup; frames are spread over the
// threads and combined in a fixed order, so the value is run-to-run identical.
struct PoolArgs {
    const float* Q;          // [n_bands][2][q_stride]
    int n_bands, n_ch, n_frames, q_stride;
    float beta_sch, beta_tch, beta_t, w_transient, jod_a, beta_jod;
    float* out;
};

__device__ __forceinline__ float pool_pow(float x, float p) { return p == 1.0f ? x : powf(x, p); }

// by the first 256 threads of a workgroup of `nthreads` >= 256 threads (all of them reach the barriers); s_part: 256 doubles
__device__ __forceinline__ void pool_jod_body(const PoolArgs& a, double* s_part, const int tid) {
    double acc = 0.0;
    if (tid < 256) {
        for (int f = tid; f < a.n_frames; f += 256) {
            float qt = 0.0f;
            for (int c = 0; c < a.n_ch; ++c) {
                const float wc = (a.n_ch == 2 && c == 1) ? a.w_transient : 1.0f;
                float qs = 0.0f;
                for (int b = 0; b < a.n_bands; ++b)
                    qs += pool_pow(fabsf(a.Q[((size_t)b * 2 + c) * a.q_stride + f] * wc), a.beta_sch);
                qs = pool_pow(qs, 1.0f / a.beta_sch);
                qt += pool_pow(qs, a.beta_tch);
            }
            qt = pool_pow(qt, 1.0f / a.beta_tch);
            acc += (double)pool_pow(qt, a.beta_t);
        }
        s_part[tid] = acc;
    }
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) s_part[tid] += s_part[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        // lp_norm(..., normalize=True): norm / N^(1/p)   (fvvdp.py:598-607)
        const float q_all = pool_pow((float)(s_part[0] / (double)a.n_frames), 1.0f / a.beta_t);
        const float sgn = a.jod_a < 0.0f ? -1.0f : 1.0f;
        a.out[0] = sgn * powf(powf(fabsf(a.jod_a), 1.0f / a.beta_jod) * q_all, a.beta_jod) + 10.0f;
    }
}

__global__ __launch_bounds__(256) void pool_jod_kernel(const PoolArgs a) {
    __shared__ double s_part[256];
    pool_jod_body(a, s_part, (int)threadIdx.x);
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

// ---- colouring of difference maps ---------------------------------------------------------------------------
// Replaces visualize_diff_map / vis_tonemap / log_luminance of the reference (pyfvvdp/visualize_diff_map.py) for the
// heat-map outputs "threshold" and "supra-threshold": the difference map indexes a colour map, which modulates a
// tone-mapped copy of the frame.  The tone curve is per frame: histogram (1024 bins) of the log-luminance of the
// context image (pyramid level 0, plane 0), cube root, cumulative sum.  Four small passes per batch:
//   colour_range_kernel  smallest positive and largest luminance of every frame
//   colour_hist_kernel   histogram of b = log(max(y, floor))
//   colour_curve_kernel  tone curve v[1024] of every frame
//   colour_map_kernel    per pixel: tone curve look-up x colour-map look-up -> fp16, planar [3][n][H][W]
#define COLOUR_BINS 1024
struct ColourArgs {
    L0Addr ctx;              // level 0: frame f at l0_frame(ctx, f), [HW][P]
    int P;
    unsigned int HW;
    unsigned int* range;     // [2][range_stride]: bit patterns of the smallest positive y / the largest y of a frame
    int range_stride;
    unsigned int* hist;      // [n][COLOUR_BINS]
    float* curve;            // [n][COLOUR_BINS]
    const float* lin01;      // [COLOUR_BINS] = torch.linspace(0, 1, 1024)
    const float* dmap;       // [n][HW]
    __half* out;             // element (c, f, p) at c*chan_stride + f*HW + p
    size_t chan_stride;
    int n_knots;
    float knots[8];
    float rgb[8][3];
    float dr;                // dynamic range of the tone-mapped image (0.6)
};

__global__ __launch_bounds__(256) void colour_range_kernel(const ColourArgs a) {
    const int f = blockIdx.y;
    const float* y = l0_frame(a.ctx, f);
    unsigned int mn = 0x7F800000u, mx = 0u;          // +inf, 0: positive floats order like their bit patterns
    for (unsigned int p = blockIdx.x * 256 + threadIdx.x; p < a.HW; p += gridDim.x * 256) {
        const float v = y[(size_t)p * a.P];
        if (v > 0.0f) {
            mn = min(mn, __float_as_uint(v));
            mx = max(mx, __float_as_uint(v));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        mn = min(mn, (unsigned int)__shfl_xor((int)mn, o, 64));
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, o, 64));
    }
    __shared__ unsigned int s_mn[4], s_mx[4];
    if ((threadIdx.x & 63) == 0) {
        s_mn[threadIdx.x >> 6] = mn;
        s_mx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&a.range[f], min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3])));
        atomicMax(&a.range[a.range_stride + f], max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3])));
    }
}

// log-luminance of a context pixel and the frame's range (log_luminance: clamp to the smallest positive value)
struct ColourFrame {
    float floor_y, b_min, b_max, span;
};
__device__ __forceinline__ ColourFrame colour_frame(const ColourArgs& a, int f) {
    ColourFrame c;
    c.floor_y = __uint_as_float(a.range[f]);
    const float top = fmaxf(__uint_as_float(a.range[a.range_stride + f]), c.floor_y);
    c.b_min = logf(c.floor_y);
    c.b_max = logf(top);
    c.span = c.b_max - c.b_min;
    return c;
}

__global__ __launch_bounds__(256) void colour_hist_kernel(const ColourArgs a) {
    __shared__ unsigned int h[COLOUR_BINS];
    const int f = blockIdx.y;
    for (int i = threadIdx.x; i < COLOUR_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const ColourFrame c = colour_frame(a, f);
    const float scale = (float)COLOUR_BINS / fmaxf(c.span, 1e-30f);
    const float* y = l0_frame(a.ctx, f);
    for (unsigned int p = blockIdx.x * 256 + threadIdx.x; p < a.HW; p += gridDim.x * 256) {
        const float b = logf(fmaxf(y[(size_t)p * a.P], c.floor_y));
        int bin = (int)((b - c.b_min) * scale);       // torch.histc: last bin closed
        bin = min(max(bin, 0), COLOUR_BINS - 1);
        atomicAdd(&h[bin], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < COLOUR_BINS; i += 256)
        if (h[i]) atomicAdd(&a.hist[(size_t)f * COLOUR_BINS + i], h[i]);
}

// vis_tonemap (visualize_diff_map.py:34-55): v = cumsum(p^(1/3) / sum(p^(1/3))) * dr + (1-dr)/2; one block per frame
__global__ __launch_bounds__(COLOUR_BINS) void colour_curve_kernel(const ColourArgs a) {
    __shared__ float s[COLOUR_BINS];
    __shared__ float s_tot;
    const int f = blockIdx.x, i = threadIdx.x;
    const float p = (float)a.hist[(size_t)f * COLOUR_BINS + i] / (float)a.HW;
    const float dy = powf(p, 1.0f / 3.0f);
    s[i] = dy;
    __syncthreads();
    // inclusive scan (Hillis-Steele): 10 steps over 1024 values
    for (int o = 1; o < COLOUR_BINS; o <<= 1) {
        const float t = (i >= o) ? s[i - o] : 0.0f;
        __syncthreads();
        s[i] += t;
        __syncthreads();
    }
    if (i == COLOUR_BINS - 1) s_tot = s[i];
    __syncthreads();
    a.curve[(size_t)f * COLOUR_BINS + i] = s[i] / s_tot * a.dr + (1.0f - a.dr) * 0.5f;
}

__global__ __launch_bounds__(256) void colour_map_kernel(const ColourArgs a) {
    __shared__ float s_v[COLOUR_BINS];
    __shared__ float s_x[COLOUR_BINS];
    __shared__ float s_knot[8];
    __shared__ float s_rgb[8][3];
    const int f = blockIdx.y;
    const ColourFrame c = colour_frame(a, f);
    for (int i = threadIdx.x; i < COLOUR_BINS; i += 256) {
        s_v[i] = a.curve[(size_t)f * COLOUR_BINS + i];
        s_x[i] = c.b_min + c.span * a.lin01[i];
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s_knot[k] = a.knots[k];
            s_rgb[k][0] = a.rgb[k][0];
            s_rgb[k][1] = a.rgb[k][1];
            s_rgb[k][2] = a.rgb[k][2];
        }
    }
    __syncthreads();
    const bool flat = c.span < a.dr;                  // low dynamic range: linear mapping (vis_tonemap :38-39)
    const float inv_step = (float)(COLOUR_BINS - 1) / fmaxf(c.span, 1e-30f);
    const float* y = l0_frame(a.ctx, f);
    const float* dm = a.dmap + (size_t)f * a.HW;
    for (unsigned int p = blockIdx.x * 256 + threadIdx.x; p < a.HW; p += gridDim.x * 256) {
        const float b = logf(fmaxf(y[(size_t)p * a.P], c.floor_y));
        float tmo;
        if (flat) {
            tmo = (b - c.b_min) / (c.span + 1e-3f) * a.dr + (1.0f - a.dr) * 0.5f;
        } else {
            // interp1 with the reference's knot search: imax = first knot >= b (clamped), imin = imax - 1
            int imax = min(max((int)ceilf((b - c.b_min) * inv_step), 0), COLOUR_BINS - 1);
            while (imax > 0 && s_x[imax - 1] >= b) --imax;
            while (imax < COLOUR_BINS - 1 && s_x[imax] < b) ++imax;
            const int imin = max(imax - 1, 0);
            float frc = (b - s_x[imin]) / (s_x[imax] - s_x[imin] + 0.000001f);
            frc = (imax == imin) ? 0.0f : fmaxf(frc, 0.0f);
            tmo = s_v[imin] * (1.0f - frc) + s_v[imax] * frc;
        }
        const float d = fminf(fmaxf(dm[p], 0.0f), 1.0f);
        int kmax = 0;
        while (kmax < a.n_knots - 1 && s_knot[kmax] < d) ++kmax;
        const int kmin = max(kmax - 1, 0);
        float fk = (d - s_knot[kmin]) / (s_knot[kmax] - s_knot[kmin] + 0.000001f);
        fk = (kmax == kmin) ? 0.0f : fmaxf(fk, 0.0f);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float col = s_rgb[kmin][ch] * (1.0f - fk) + s_rgb[kmax][ch] * fk;
            const float v = fminf(fmaxf(col * tmo, 0.0f), 1.0f);
            a.out[(size_t)ch * a.chan_stride + (size_t)f * a.HW + p] = __float2half_rn(v);
        }
    }
}

// Streaming write of n4 float4 into each of q0 and q1 at once (even workgroups -> q0, odd -> q1): the probe of choose_level0.  Two ranges
// of different classes of physical memory take ~7 TB/s together, two of the same class ~5.5.  Grid-stride, 16 B per lane, non-temporal.
__global__ __launch_bounds__(256) void stream_write_probe_kernel(float4* __restrict__ q0, float4* __restrict__ q1, size_t n4, float v) {
    float4* q = (blockIdx.x & 1) ? q1 : q0;
    size_t i = (size_t)(blockIdx.x >> 1) * 1024 + threadIdx.x;
    const size_t stride = (size_t)(gridDim.x >> 1) * 1024;
    for (; i + 768 < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v4f{v, v + (float)u, v, v}, reinterpret_cast<v4f*>(q + i + u * 256));
    }
}
