// Front end of the one K1 -> K2 fusion variant that was never costed on silicon: (strip, chunk)-
// persistent single-wave workgroups walk the FRAMES of their tile in order and RE-COMPUTE the four temporal channels of
// every level-0 pixel they need from the uint8 source frames (8-frame window re-read through L2 / Infinity Cache), instead
// of reading a level 0 that a separate temporal kernel wrote.  This microbenchmark runs ONLY that front end -- unpack,
// code -> luminance table in LDS, RGB -> Y, 8-tap temporal FIR of both streams -- and folds the result into a checksum
// (no level 0 is written, no pyramid is computed).  If the front end alone costs about as much as K1 + the hand-off it is
// meant to remove, the fused kernel (front end + K2b's VALU-bound body in one wave) cannot win.
//
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/fuse_front.hip -o build_variants/fuse_front
//   build_variants/fuse_front [rows_per_chunk=64] [halo_rows=14] [frames=60]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

struct Args {
    const unsigned char* src[2];   // [3][N][H][W] uint8 per stream
    const float* lut;              // [256] code -> luminance
    float w[3];
    float taps[8][2];
    int W, H, N, rows, halo, strips, chunks;
    float* out;                    // one checksum per workgroup
};

// lane owns 4 consecutive pixels (one dword per channel, frame and stream); a wave covers 256 columns
__global__ __launch_bounds__(64, 4) void front_recompute(const Args a) {
    __shared__ float lutw[3][256];
    for (int i = threadIdx.x; i < 256; i += 64) {
        const float l = a.lut[i];
        lutw[0][i] = l * a.w[0]; lutw[1][i] = l * a.w[1]; lutw[2][i] = l * a.w[2];
    }
    __syncthreads();
    const int lane = threadIdx.x;
    const int strip = blockIdx.x % a.strips, chunk = blockIdx.x / a.strips;
    const int x0 = min(strip * 256 + lane * 4, a.W - 4);
    const int r0 = chunk * a.rows - a.halo / 2;
    const size_t HW = (size_t)a.W * a.H, CS = HW * a.N;
    v2f acc = v2f{0.0f, 0.0f};
    for (int f = 0; f < a.N; ++f) {
        for (int r = 0; r < a.rows + a.halo; ++r) {
            const int y = min(max(r0 + r, 0), a.H - 1);
            const size_t po = (size_t)y * a.W + x0;
            v2f s[4], t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = t[i] = v2f{0.0f, 0.0f};
#pragma unroll
            for (int k = 7; k >= 0; --k) {
                const int ff = max(f - k, 0);
                unsigned int wd[2][3];
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        wd[st][c] = *reinterpret_cast<const unsigned int*>(a.src[st] + c * CS + (size_t)ff * HW + po);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float L[2];
#pragma unroll
                    for (int st = 0; st < 2; ++st)
                        L[st] = (lutw[0][(wd[st][0] >> (8 * i)) & 255] + lutw[1][(wd[st][1] >> (8 * i)) & 255]) + lutw[2][(wd[st][2] >> (8 * i)) & 255];
                    const v2f p = v2f{L[0], L[1]};
                    s[i] = __builtin_elementwise_fma(p, v2f{a.taps[k][0], a.taps[k][0]}, s[i]);
                    t[i] = __builtin_elementwise_fma(p, v2f{a.taps[k][1], a.taps[k][1]}, t[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += s[i] * 0.5f + t[i];
        }
    }
    float v = acc.x + acc.y;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) a.out[blockIdx.x] = v;
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 64, halo = argc > 2 ? atoi(argv[2]) : 14, N = argc > 3 ? atoi(argv[3]) : 60;
    const int W = 3840, H = 2160;
    Args a;
    const size_t bytes = (size_t)3 * N * W * H;
    unsigned char* d[2];
    for (int s = 0; s < 2; ++s) {
        CHECK(hipMalloc(&d[s], bytes));
        std::vector<unsigned char> h(bytes);
        unsigned int x = 12345u + s;
        for (size_t i = 0; i < bytes; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned char)(x >> 24); }
        CHECK(hipMemcpy(d[s], h.data(), bytes, hipMemcpyHostToDevice));
        a.src[s] = d[s];
    }
    float hl[256];
    for (int i = 0; i < 256; ++i) hl[i] = 0.2f + 199.8f * (float)i / 255.0f;
    float* dl; CHECK(hipMalloc(&dl, sizeof(hl))); CHECK(hipMemcpy(dl, hl, sizeof(hl), hipMemcpyHostToDevice));
    a.lut = dl; a.w[0] = 0.2126f; a.w[1] = 0.7152f; a.w[2] = 0.0722f;
    for (int k = 0; k < 8; ++k) { a.taps[k][0] = 0.125f; a.taps[k][1] = (k & 1) ? -0.1f : 0.1f; }
    a.W = W; a.H = H; a.N = N; a.rows = rows; a.halo = halo;
    a.strips = W / 256; a.chunks = (H + rows - 1) / rows;
    const int nblk = a.strips * a.chunks;
    CHECK(hipMalloc(&a.out, nblk * sizeof(float)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(front_recompute, dim3(nblk), dim3(64), 0, 0, a);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    const double px = (double)W * H * N * (double)(rows + halo) / rows;
    printf("front end only, rows %d + halo %d, %d workgroups: %.3f ms per %d frames = %.1f us/frame (owned pixels), %.2f x pixels incl. halo, "
           "%.1f GB/s of uint8 window reads\n", rows, halo, nblk, best, N, best / N * 1e3, (double)(rows + halo) / rows, px * 48.0 / (best * 1e-3) / 1e9);
    return 0;
}
