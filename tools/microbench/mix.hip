// HBM read:write mix micro-benchmark (not part of the product): every thread streams R float4 reads per float4 written,
// fully coalesced, contiguous 1-KiB runs per wave.  Shows what the memory system gives for the pyramid kernels' 4:1
// (level i -> i+1) and 16:1 (level i -> i+2) mixes when the access pattern itself is ideal.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mix.hip -o build_variants/mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float vf4 __attribute__((ext_vector_type(4)));

// grid-stride over "units": unit u reads float4 [u*R*64*K .. ) and writes float4 [u*64*K ..): K consecutive 1-KiB runs per wave
template <int R, int NT, int WR>
__global__ __launch_bounds__(256) void mix(const vf4* __restrict__ in, vf4* __restrict__ out, size_t n_out4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_out4; i += stride) {
        // wave-contiguous: the wave's 64 lanes read R runs of 1 KiB each
        const size_t wave = i >> 6, lane = i & 63;
        vf4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = in[(wave * R + r) * 64 + lane];
        vf4 s = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) s += v[r];
        if constexpr (WR == 1) {
            if constexpr (NT) __builtin_nontemporal_store(s, out + i); else out[i] = s;
        } else {
            if (s.x == 123.456f) out[i] = s;
        }
    }
}

// write-heavy mixes (the temporal kernel: 6 B read, 16 B written per pixel): 1 float4 read, WN float4 written per thread
template <int WN>
__global__ __launch_bounds__(256) void wmix(const vf4* __restrict__ in, vf4* __restrict__ out, size_t n_in4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_in4; i += stride) {
        const size_t wave = i >> 6, lane = i & 63;
        const vf4 v = in[i];
#pragma unroll
        for (int r = 0; r < WN; ++r) __builtin_nontemporal_store(v, out + (wave * WN + r) * 64 + lane);
    }
}

template <int WN>
static void runw(const vf4* in, vf4* out, size_t n_out4_total) {
    const size_t n_in4 = n_out4_total / WN;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {4096, 16384}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((wmix<WN>), dim3(blocks), dim3(256), 0, 0, in, out, n_in4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double rd = (double)n_in4 * 16, wr = (double)n_in4 * WN * 16;
        printf("R:W =  1:%d nt store     blocks %5d: %.3f ms  read %.2f + write %.2f = %.2f TB/s\n", WN, blocks, best, rd / best / 1e9,
               wr / best / 1e9, (rd + wr) / best / 1e9);
    }
}

template <int R, int NT, int WR>
static void run(const vf4* in, vf4* out, size_t n_in4, const char* tag) {
    const size_t n_out4 = n_in4 / R;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {4096, 16384}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((mix<R, NT, WR>), dim3(blocks), dim3(256), 0, 0, in, out, n_out4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double rd = (double)n_out4 * R * 16, wr = WR ? (double)n_out4 * 16 : 0;
        printf("R:W = %2d:%d %-12s blocks %5d: %.3f ms  read %.2f + write %.2f = %.2f TB/s\n", R, WR, tag, blocks, best, rd / best / 1e9,
               wr / best / 1e9, (rd + wr) / best / 1e9);
    }
}

int main() {
    const size_t n4 = (size_t)3840 * 2160 * 60;      // 7.96 GB read
    vf4 *p, *q;
    CK(hipMalloc(&p, n4 * 16)); CK(hipMalloc(&q, n4 * 16));
    CK(hipMemset(p, 1, n4 * 16));
    run<4, 0, 0>(p, q, n4, "read only");
    run<1, 1, 1>(p, q, n4, "nt store");
    run<2, 1, 1>(p, q, n4, "nt store");
    run<4, 1, 1>(p, q, n4, "nt store");
    run<4, 0, 1>(p, q, n4, "plain store");
    run<8, 1, 1>(p, q, n4, "nt store");
    run<16, 1, 1>(p, q, n4, "nt store");
    run<16, 0, 1>(p, q, n4, "plain store");
    runw<2>(p, q, n4);
    runw<3>(p, q, n4);
    runw<4>(p, q, n4);
    runw<8>(p, q, n4);
    return 0;
}
