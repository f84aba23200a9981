// Round 5: does the price of mixing a write stream into a read stream depend on whether the two ranges lie in the same class of the box's
// physical memory (profiles/r05_k1_mode.md section 3: two classes; writes into one 5.5 TB/s, into both at once 7.0)?
//   N ranges of 4 GB (chunk-mapped and hipMalloc in turn) -> pairwise write probe = their classes -> for pairs of the same / of different
//   classes: read A while writing B at 1:16 (the pyramid pass), 16:6 (the temporal kernel: 6 B read per 16 B written) and 1:1 (a copy).
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/rw_classes.hip -o build_variants/rw_classes
//   run:    build_variants/rw_classes [N=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// workgroups [0, n_rd) read `src` (n4 float4), the others write `dst`; each stream is spread evenly over its workgroups
__global__ __launch_bounds__(256) void read_write(const v4f* __restrict__ src, size_t n4_src, v4f* __restrict__ dst, size_t n4_dst, int n_rd, float* sink) {
    const int b = blockIdx.x;
    if (b < n_rd) {
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        const size_t stride = (size_t)n_rd * 1024;
        for (size_t i = (size_t)b * 1024 + threadIdx.x; i + 768 < n4_src; i += stride) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += __builtin_nontemporal_load(src + i + u * 256);
        }
        if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[0] = acc.x;
    } else {
        const int n_wr = gridDim.x - n_rd;
        const size_t stride = (size_t)n_wr * 1024;
        for (size_t i = (size_t)(b - n_rd) * 1024 + threadIdx.x; i + 768 < n4_dst; i += stride) {
#pragma unroll
            for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v4f{1.f, (float)u, 2.f, 3.f}, dst + i + u * 256);
        }
    }
}

struct Buf { void* ptr; int kind; };
static Buf alloc_buf(size_t bytes, int kind) {
    Buf b; b.ptr = nullptr; b.kind = kind;
    if (!kind) { CK(hipMalloc(&b.ptr, bytes)); return b; }
    const size_t chunk = (size_t)32 << 20;
    int dev = 0; CK(hipGetDevice(&dev));
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    CK(hipMemAddressReserve(&b.ptr, bytes, gran, nullptr, 0));
    for (size_t i = 0; i < bytes / chunk; ++i) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char*)b.ptr + i * chunk, chunk, 0, h, 0));
    }
    hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc)); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.ptr, bytes, &acc, 1));
    return b;
}

template <typename F> static double time_us(F f, int reps) {
    static hipEvent_t e0 = nullptr, e1 = nullptr;
    if (!e0) { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
    std::vector<double> v;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.push_back(ms * 1e3);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8;
    const size_t bytes = (size_t)4 << 30, n4 = bytes / 16;
    std::vector<Buf> bufs;
    for (int k = 0; k < N; ++k) bufs.push_back(alloc_buf(bytes, k % 2 == 0 ? 1 : 0));
    float* sink = nullptr; CK(hipMalloc(&sink, 64));
    const int G = 32768;
    // both streams of n4 float4 each
    auto rate = [&](const void* src, size_t n_src, void* dst, size_t n_dst, int n_rd) {
        const double us = time_us([&] { hipLaunchKernelGGL(read_write, dim3(G), dim3(256), 0, 0, (const v4f*)src, n_src, (v4f*)dst, n_dst, n_rd, sink); }, 3);
        return (double)(n_src + n_dst) * 16.0 / us * 1e-6;          // TB/s
    };
    for (auto& b : bufs) (void)rate(b.ptr, 0, b.ptr, n4, 0);         // first touch
    // classes: write two ranges at once, each from its own stream -- a pair of different classes takes ~7 TB/s, of one class ~5.5
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    auto pair_write = [&](int i, int j) {
        double best = 0.0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0));
            CK(hipStreamWaitEvent(s1, e0, 0));
            hipLaunchKernelGGL(read_write, dim3(G / 2), dim3(256), 0, s0, (const v4f*)nullptr, (size_t)0, (v4f*)bufs[i].ptr, n4, 0, sink);
            hipLaunchKernelGGL(read_write, dim3(G / 2), dim3(256), 0, s1, (const v4f*)nullptr, (size_t)0, (v4f*)bufs[j].ptr, n4, 0, sink);
            CK(hipEventRecord(e1, s1));
            CK(hipStreamWaitEvent(s0, e1, 0));
            CK(hipEventRecord(e2, s0));
            CK(hipEventSynchronize(e2));
            float ms; CK(hipEventElapsedTime(&ms, e0, e2));
            best = std::max(best, 2.0 * (double)bytes / (ms * 1e-3) / 1e12);
        }
        return best;
    };
    std::vector<int> cls(N, 0);
    printf("write rate together with range 0 [TB/s]:");
    for (int j = 1; j < N; ++j) { const double r = pair_write(0, j); cls[j] = r >= 6.6 ? 1 : 0; printf("  %d:%.2f", j, r); }
    printf("\nclasses (0 = that of range 0):");
    for (int j = 0; j < N; ++j) printf(" %d", cls[j]);
    printf("\n\n%-34s %10s %10s %10s %10s   [TB/s of both streams; workgroups split in the ratio of the bytes]\n", "read A + write B", "read only", "1:16", "6:16", "1:1");
    int shown_same = 0, shown_diff = 0;
    for (int i = 0; i < N && (shown_same < 3 || shown_diff < 3); ++i)
        for (int j = 0; j < N; ++j) {
            if (i == j) continue;
            const bool same = cls[i] == cls[j];
            if ((same && shown_same >= 3) || (!same && shown_diff >= 3)) continue;
            (same ? shown_same : shown_diff) += 1;
            const double r0 = rate(bufs[i].ptr, n4, bufs[j].ptr, 0, G);
            const double r1 = rate(bufs[i].ptr, n4, bufs[j].ptr, n4 / 16, G - G / 17);
            const double r2 = rate(bufs[i].ptr, n4 * 6 / 16, bufs[j].ptr, n4, G * 6 / 22);
            const double r3 = rate(bufs[i].ptr, n4, bufs[j].ptr, n4, G / 2);
            printf("A=%d (class %d) B=%d (class %d) %-6s %10.2f %10.2f %10.2f %10.2f\n", i, cls[i], j, cls[j], same ? "same" : "differ", r0, r1, r2, r3);
        }
    // one range as both source and destination (its two halves)
    for (int i = 0; i < 2; ++i) {
        const char* a = (const char*)bufs[i].ptr; char* b = (char*)bufs[i].ptr + bytes / 2;
        const size_t h4 = n4 / 2;
        printf("A = B = %d (halves of one range)          %10.2f %10.2f %10.2f %10.2f\n", i, rate(a, h4, b, 0, G), rate(a, h4, b, h4 / 16, G - G / 17),
               rate(a, h4 * 6 / 16, b, h4, G * 6 / 22), rate(a, h4, b, h4, G / 2));
    }
    return 0;
}
