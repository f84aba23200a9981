// Round 4: do the streaming ceilings of tools/microbench/membw.hip depend on whether a buffer is one physically contiguous
// range (hipMalloc on a box with free memory) or mapped from physical chunks through the virtual-memory API?  (Not part of the
// product.)  usage: chunks [chunk_MB ...]   (0 = hipMalloc)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_f4(const float4* __restrict__ p, size_t n4, float* out) {
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
__global__ __launch_bounds__(256) void write_f4(float4* __restrict__ q, size_t n4, float v) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 1024;
    for (; i + 768 < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) q[i + u * 256] = make_float4(v, v + u, v, v);
    }
}
__global__ __launch_bounds__(256) void copy_f4(const float4* __restrict__ p, float4* __restrict__ q, size_t n4) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 1024;
    for (; i + 768 < n4; i += stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[i + u * 256] = v[u];
    }
}
// the temporal kernel's mix: 6 B read (here: 8 B, one float2 of a separate source) and 16 B written per element
__global__ __launch_bounds__(256) void expand_f2_f4(const float2* __restrict__ p, float4* __restrict__ q, size_t n) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 1024;
    for (; i + 768 < n; i += stride) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[i + u * 256] = make_float4(v[u].x, v[u].y, v[u].x, v[u].y);
    }
}

struct Buf { void* ptr; size_t size; std::vector<hipMemGenericAllocationHandle_t> h; bool vmm; };
static Buf alloc(size_t bytes, size_t chunk) {
    Buf b; b.vmm = chunk != 0; b.size = bytes; b.ptr = nullptr;
    if (!chunk) { CK(hipMalloc(&b.ptr, bytes)); return b; }
    int dev = 0; CK(hipGetDevice(&dev));
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran;
    b.size = (bytes + chunk - 1) / chunk * chunk;
    CK(hipMemAddressReserve(&b.ptr, b.size, gran, nullptr, 0));
    for (size_t i = 0; i < b.size / chunk; ++i) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); b.h.push_back(h);
        CK(hipMemMap((char*)b.ptr + i * chunk, chunk, 0, h, 0));
    }
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc)); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.ptr, b.size, &acc, 1));
    return b;
}
static void release(Buf& b) {
    if (!b.vmm) { CK(hipFree(b.ptr)); return; }
    CK(hipMemUnmap(b.ptr, b.size));
    for (auto h : b.h) CK(hipMemRelease(h));
    CK(hipMemAddressFree(b.ptr, b.size));
}
template <typename F> static double timeit(F f, int reps = 6) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e-3;
}
int main(int argc, char** argv) {
    const size_t n4 = (size_t)3840 * 2160 * 60;       // 7.96 GB of float4
    std::vector<size_t> chunks;
    for (int i = 1; i < argc; ++i) chunks.push_back((size_t)atoll(argv[i]) << 20);
    if (chunks.empty()) chunks = {0, (size_t)32 << 20};
    float* out; CK(hipMalloc(&out, 64));
    for (int rep = 0; rep < 2; ++rep)
        for (size_t ch : chunks) {
            Buf p = alloc(n4 * 16, ch), q = alloc(n4 * 16, ch), s = alloc(n4 * 8, 0);
            CK(hipMemset(p.ptr, 1, n4 * 16)); CK(hipMemset(q.ptr, 1, n4 * 16)); CK(hipMemset(s.ptr, 1, n4 * 8));
            const double gb = n4 * 16 / 1e9;
            const int blocks = 32768;
            const double tr = timeit([&] { hipLaunchKernelGGL(read_f4, dim3(blocks), dim3(256), 0, 0, (const float4*)p.ptr, n4, out); });
            const double tw = timeit([&] { hipLaunchKernelGGL(write_f4, dim3(blocks), dim3(256), 0, 0, (float4*)q.ptr, n4, 1.0f); });
            const double tc = timeit([&] { hipLaunchKernelGGL(copy_f4, dim3(blocks), dim3(256), 0, 0, (const float4*)p.ptr, (float4*)q.ptr, n4); });
            const double te = timeit([&] { hipLaunchKernelGGL(expand_f2_f4, dim3(blocks), dim3(256), 0, 0, (const float2*)s.ptr, (float4*)q.ptr, n4); });
            printf("%-22s read %.2f | write %.2f | copy %.2f (r+w) | 8 B read (hipMalloc source) + 16 B written %.2f TB/s\n",
                   ch ? (std::to_string(ch >> 20) + " MB chunks").c_str() : "hipMalloc", gb / tr / 1e3, gb / tw / 1e3, 2 * gb / tc / 1e3, 1.5 * gb / te / 1e3);
            fflush(stdout);
            release(p); release(q); release(s);
        }
    return 0;
}
