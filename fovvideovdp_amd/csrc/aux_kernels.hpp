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

