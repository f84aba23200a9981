// Does a producer->consumer hand-off through a reused scratch buffer stay in the 256 MiB Infinity Cache (MALL)?
// write kernel fills S bytes, read kernel consumes them; effective bandwidth vs S (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void wr(float4* q, size_t n4, float v) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 1024;
    for (; i + 768 < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) q[i + u * 256] = make_float4(v, v + u, v, v);
    }
}
__global__ __launch_bounds__(256) void rd(const float4* p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 1024;
    float acc = 0.f;
    for (; i + 768 < n4; i += stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// streaming side traffic (other data flowing through the memory system at the same time)
__global__ __launch_bounds__(256) void rd_stream(const float4* p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 1024;
    float acc = 0.f;
    for (; i + 768 < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { float4 v = p[i + u * 256]; acc += v.x + v.w; }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    float* out; CK(hipMalloc(&out, 64));
    float4* big; const size_t bigN = (size_t)1 << 28;   // 4 GiB of float4
    CK(hipMalloc(&big, bigN * 16)); CK(hipMemset(big, 0, bigN * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (size_t mb : {32, 64, 96, 128, 160, 192, 256, 384, 1024, 2048}) {
        const size_t n4 = mb * 1024 * 1024 / 16;
        float4* buf; CK(hipMalloc(&buf, n4 * 16));
        const int blocks = 4096;
        const int reps = (int)(16384 / mb) + 2;
        for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, buf, n4, 1.f); hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, buf, n4, out); }
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, buf, n4, (float)r);
            hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, buf, n4, out);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = 2.0 * n4 * 16 * reps;
        printf("scratch %5zu MiB: write+read hand-off %.2f TB/s (%.1f us per pair)\n", mb, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / reps);
        // same with 1:1 streaming traffic in between (the pyramid kernel also streams other data)
        CK(hipEventRecord(e0));
        size_t off = 0;
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, buf, n4, (float)r);
            hipLaunchKernelGGL(rd_stream, dim3(blocks), dim3(256), 0, 0, big + off, n4 / 2, out);
            off = (off + n4 / 2) % (bigN - n4);
            hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, buf, n4, out);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("                  with 0.5x streaming reads in between: %.2f TB/s (all bytes)\n", 2.5 * n4 * 16 * reps / (ms * 1e-3) / 1e12);
        CK(hipFree(buf));
    }
    return 0;
}
