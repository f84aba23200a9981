// Access-pattern ceiling of the pyramid kernels (not part of the product): single-wave workgroups walk down strips of a
// [N][H][W] float4 image, 2 x float4 per lane per row, 2 rows per step, one step prefetched, XCD-aware work order.
// Variants (compile-time, so the loop body is straight-line code): strip pitch (owned lanes), halo rows re-read per
// chunk, what is written (level i+1: 1 float4 per owned lane per step; level i+2: 1 float4 per owned lane PAIR every
// second step; nothing) and how (branchy global store vs unconditional buffer store with out-of-range offsets).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/strips.hip -o build_variants/strips
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef int vi4 __attribute__((ext_vector_type(4)));

struct A { const float4* img; float4* out1; float4* out2; float* sink; int W, H, n_strips, n_chunks, cr, active, hl, pre, valu; };

// WMODE 0 none, 1 level i+1 branchy, 2 level i+1 straight-line, 3 level i+2 branchy, 4 level i+2 straight-line
template <int WMODE, int DEPTH>
__global__ __launch_bounds__(64) void strips(const A a) {
    int bid;
    { const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7; bid = x * q8 + min(x, r8) + (blockIdx.x >> 3); }
    const int strip = bid % a.n_strips;
    const int chunk = (bid / a.n_strips) % a.n_chunks;
    const int frame = bid / (a.n_strips * a.n_chunks);
    const int lane = threadIdx.x;
    const int J = strip * a.active + lane - a.hl;
    const int W = a.W, H = a.H, Wc = W / 2, Hc = H / 2;
    const int x0 = min(max(2 * J, 0), W - 2);
    const float4* base = a.img + (size_t)frame * W * H;
    const int c0 = chunk * a.cr, c1 = min(c0 + a.cr, Hc);
    const bool own = lane >= a.hl && lane < a.hl + a.active && J < Wc;
    float acc = 0.f;
    auto row = [&](int r) { return base + (size_t)min(max(r, 0), H - 1) * W + x0; };
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(a.out1 + (size_t)frame * Hc * Wc, 0, Hc * Wc * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(a.out2 + (size_t)frame * (Hc / 2) * (Wc / 2), 0, (Hc / 2) * (Wc / 2) * 16, 0x00020000);
    int r = 2 * c0 - a.pre;                                  // halo rows of the prologue
    float4 p[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { p[d][0] = row(r + 2 * d)[0]; p[d][1] = row(r + 2 * d)[1]; p[d][2] = row(r + 2 * d + 1)[0]; p[d][3] = row(r + 2 * d + 1)[1]; }
    for (; r < 2 * c1; r += 2) {
        const float4 q0 = p[0][0], q1 = p[0][1], q2 = p[0][2], q3 = p[0][3];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) { p[d][0] = p[d + 1][0]; p[d][1] = p[d + 1][1]; p[d][2] = p[d + 1][2]; p[d][3] = p[d + 1][3]; }
        const int rn = r + 2 * DEPTH;
        p[DEPTH - 1][0] = row(rn)[0]; p[DEPTH - 1][1] = row(rn)[1]; p[DEPTH - 1][2] = row(rn + 1)[0]; p[DEPTH - 1][3] = row(rn + 1)[1];
        float4 v = make_float4(q0.x + q1.y + q2.z + q3.w, q0.y + q1.z + q2.w + q3.x, q0.z + q1.w + q2.x + q3.y, q0.w + q1.x + q2.y + q3.z);
        for (int i = 0; i < a.valu; ++i) { v.x = fmaf(v.x, 1.0001f, v.y); v.y = fmaf(v.y, 0.9999f, v.z); v.z = fmaf(v.z, 1.0002f, v.w); v.w = fmaf(v.w, 0.9998f, v.x); }
        acc += v.x + v.y + v.z + v.w;
        const int c = r >> 1;
        const bool wr = r >= 2 * c0 && own;
        if constexpr (WMODE == 1) { if (wr) __builtin_nontemporal_store(vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<vf4*>(a.out1 + ((size_t)frame * Hc + c) * Wc + J)); }
        if constexpr (WMODE == 2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vi4, vf4{v.x, v.y, v.z, v.w}), rs1, wr ? (unsigned)(c * Wc + J) * 16u : 0xFFFFFFFFu, 0, 2);
        if constexpr (WMODE == 3) { if (wr && (c & 1) == 0 && (lane & 1) == 0) __builtin_nontemporal_store(vf4{v.x, v.y, v.z, v.w}, reinterpret_cast<vf4*>(a.out2 + ((size_t)frame * (Hc / 2) + (c >> 1)) * (Wc / 2) + (J >> 1))); }
        if constexpr (WMODE == 4) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vi4, vf4{v.x, v.y, v.z, v.w}), rs2,
                (wr && (c & 1) == 0 && (lane & 1) == 0) ? (unsigned)((c >> 1) * (Wc / 2) + (J >> 1)) * 16u : 0xFFFFFFFFu, 0, 2);
    }
    if (acc == 123.456f) a.sink[0] = acc;
}

typedef void (*KFn)(const A);
int main(int argc, char** argv) {
    const int W = 3840, H = 2160, N = 60;
    const size_t n4 = (size_t)W * H * N;
    float4 *p, *o1, *o2; float* sink;
    CK(hipMalloc(&p, n4 * 16)); CK(hipMalloc(&o1, n4 * 4)); CK(hipMalloc(&o2, n4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(p, 1, n4 * 16));
    struct V { const char* name; KFn fn; int active, hl, pre, cr, valu; };
    const V vs[] = {
        {"pitch 60, no write, cr 30", strips<0, 1>, 60, 2, 4, 30, 0},
        {"pitch 60, no write, cr 30, prefetch 2 steps", strips<0, 2>, 60, 2, 4, 30, 0},
        {"pitch 60, L+1 branchy store (round 1), cr 30", strips<1, 1>, 60, 2, 4, 30, 0},
        {"pitch 60, L+1 straight-line store, cr 30", strips<2, 1>, 60, 2, 4, 30, 0},
        {"pitch 60, L+1 straight-line store, cr 30, prefetch 2 steps", strips<2, 2>, 60, 2, 4, 30, 0},
        {"pitch 60, L+1 straight-line store, cr 60", strips<2, 1>, 60, 2, 4, 60, 0},
        {"pitch 52, no write, cr 60, pre 12", strips<0, 1>, 52, 8, 12, 60, 0},
        {"pitch 52, L+2 branchy store, cr 60, pre 12", strips<3, 1>, 52, 8, 12, 60, 0},
        {"pitch 52, L+2 straight-line store, cr 60, pre 12", strips<4, 1>, 52, 8, 12, 60, 0},
        {"pitch 52, L+2 straight-line store, cr 60, pre 12, prefetch 2", strips<4, 2>, 52, 8, 12, 60, 0},
        {"pitch 52, L+2 straight-line store, cr 120, pre 12", strips<4, 1>, 52, 8, 12, 120, 0},
        {"pitch 52, L+2 straight-line, cr 60, +256 fma/step", strips<4, 1>, 52, 8, 12, 60, 64},
        {"pitch 52, L+2 straight-line, cr 60, +384 fma/step", strips<4, 1>, 52, 8, 12, 60, 96},
        {"pitch 60, L+1 straight-line, cr 30, +256 fma/step", strips<2, 1>, 60, 2, 4, 30, 64},
    };
    for (const V& v : vs) {
        A a{p, o1, o2, sink, W, H, (W / 2 + v.active - 1) / v.active, (H / 2 + v.cr - 1) / v.cr, v.cr, v.active, v.hl, v.pre, v.valu};
        const int grid = a.n_strips * a.n_chunks * N;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        double best = 1e9, sum = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(v.fn, dim3(grid), dim3(64), 0, 0, a);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-62s: %6.2f us/frame best, %6.2f avg  (level-0 read alone = %.2f TB/s)\n", v.name, best * 1e3 / N, sum / 5 * 1e3 / N,
               (double)W * H * 16 * N / (best * 1e-3) / 1e12);
    }
    return 0;
}
