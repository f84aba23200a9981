// HBM streaming micro-benchmarks used to find the practical ceilings quoted in DESIGN.md (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#include <cstring>
#include <string>

template <int UNROLL>
__global__ __launch_bounds__(256) void read_f4(const float4* __restrict__ p, size_t n4, float* out) {
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    float acc = 0.f;
    for (; i + 256 * (UNROLL - 1) < n4; i += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void copy_f4(const float4* __restrict__ p, float4* __restrict__ q, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (; i + 256 * (UNROLL - 1) < n4; i += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) q[i + u * 256] = v[u];
    }
}

// the band kernel's read pattern: single-wave workgroups, each walks down `rows` rows of a [H][W] float4 image,
// reading 2 x float4 per lane per row (lane l: pixels 2l, 2l+1 of a 128-pixel strip), 2 rows per step, prefetch 1.
__device__ size_t g_skew = 0;
template <int ACTIVE, int HALO_L, int NT, int WPB = 1>
__global__ __launch_bounds__(64 * WPB) void read_strips(const float4* __restrict__ img, int W, int H, int n_strips, int chunk_rows,
                                                  int n_chunks, float4* __restrict__ coarse, float* out) {
    int bid;
    {
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, x = blockIdx.x & 7;
        bid = x * q8 + min(x, r8) + (blockIdx.x >> 3);
        bid = bid * WPB + (int)(threadIdx.x >> 6);        // WPB adjacent strips per workgroup, kept in step by a barrier
    }
    const int strip = bid % n_strips;
    const int chunk = (bid / n_strips) % n_chunks;
    const int frame = bid / (n_strips * n_chunks);
    const int lane = threadIdx.x & 63;
    const int x0 = min(max(strip * 2 * ACTIVE + 2 * (lane - HALO_L), 0), W - 2);
    const float4* base = img + (size_t)frame * ((size_t)W * H + g_skew);
    const int r0 = chunk * chunk_rows, r1 = min(r0 + chunk_rows, H);
    float acc = 0.f;
    auto LD = [&](size_t i) -> float4 {
        typedef float vf4 __attribute__((ext_vector_type(4)));
        if constexpr (NT & 1) { const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(base + i)); return make_float4(t.x, t.y, t.z, t.w); }
        else return base[i];
    };
    float4 a = LD((size_t)r0 * W + x0), b = LD((size_t)r0 * W + x0 + 1);
    float4 c = LD((size_t)(r0 + 1) * W + x0), d = LD((size_t)(r0 + 1) * W + x0 + 1);
    for (int r = r0; r < r1; r += 2) {
        const float4 a0 = a, b0 = b, c0 = c, d0 = d;
        if constexpr (WPB > 1) __syncthreads();
        const int rn = min(r + 2, H - 2);
        a = LD((size_t)rn * W + x0); b = LD((size_t)rn * W + x0 + 1);
        c = LD((size_t)(rn + 1) * W + x0); d = LD((size_t)(rn + 1) * W + x0 + 1);
        acc += a0.x + b0.y + c0.z + d0.w;
        if (coarse && lane >= HALO_L && lane < HALO_L + ACTIVE) {
            const int Wc = W / 2;
            const int J = strip * ACTIVE + lane - HALO_L;
            if (J < Wc) {
                float4* dst = (NT & 4) ? coarse + (((size_t)frame * n_strips + strip) * (H / 2) + r / 2) * 64 + lane
                                       : coarse + ((size_t)frame * (H / 2) + r / 2) * Wc + J;
                const float4 val = make_float4(a0.x, b0.x, c0.x, d0.x);
                typedef float vf4 __attribute__((ext_vector_type(4)));
                if constexpr (NT & 2) __builtin_nontemporal_store(vf4{val.x, val.y, val.z, val.w}, reinterpret_cast<vf4*>(dst));
                else *dst = val;
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
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
template <typename F>
static double timeit(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps * 1e-3;
}

int main() {
    const int W = 3840, H = 2160, N = 60;
    const size_t n4 = (size_t)W * H * N;            // 7.96 GB
    float4 *p, *q; float* out;
    const size_t chunk = getenv("CHUNK_MB") ? ((size_t)atoll(getenv("CHUNK_MB")) << 20) : 0;      // round 4: 0 = hipMalloc, else VMM chunks
    Buf bp = alloc(n4 * 16, chunk), bq = alloc(n4 * 16, chunk);
    p = (float4*)bp.ptr; q = (float4*)bq.ptr;
    printf("buffers: %s\n", chunk ? "mapped from physical chunks (CHUNK_MB)" : "hipMalloc");
    CK(hipMalloc(&out, 64));
    CK(hipMemset(p, 1, n4 * 16));
    const double gb = n4 * 16 / 1e9;
    for (int blocks : {2048, 4096, 8192, 32768, 131072}) {
        double t = timeit([&] { hipLaunchKernelGGL(read_f4<4>, dim3(blocks), dim3(256), 0, 0, p, n4, out); });
        printf("read_f4<4>  blocks %6d: %.2f TB/s\n", blocks, gb / t / 1e3);
    }
    for (int blocks : {4096, 32768}) {
        double t = timeit([&] { hipLaunchKernelGGL(read_f4<8>, dim3(blocks), dim3(256), 0, 0, p, n4, out); });
        printf("read_f4<8>  blocks %6d: %.2f TB/s\n", blocks, gb / t / 1e3);
    }
    for (int blocks : {4096, 32768, 131072}) {
        double t = timeit([&] { hipLaunchKernelGGL(copy_f4<4>, dim3(blocks), dim3(256), 0, 0, p, q, n4); });
        printf("copy_f4<4>  blocks %6d: %.2f TB/s (r+w)\n", blocks, 2 * gb / t / 1e3);
    }
    for (size_t skew : {(size_t)0, (size_t)16, (size_t)1040, (size_t)65552}) {     // float4 units
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_skew), &skew, sizeof(skew)));
        const int chunk_rows = 1080, n_chunks = 2, n_strips = 32, grid = n_strips * n_chunks * 59;
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            double t = timeit([&] { hipLaunchKernelGGL((read_strips<60, 0, 2>), dim3(grid), dim3(64), 0, 0, p, W, H, n_strips, chunk_rows, n_chunks, q, out); });
            best = t < best ? t : best;
        }
        printf("frame skew %6zu float4: read_strips + nt coarse write: %.2f TB/s alg (59 frames)\n", skew, 1.25 * gb * 59 / 60 / best / 1e3);
        {
            double b4 = 1e9, b2 = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                double t = timeit([&] { hipLaunchKernelGGL((read_strips<60, 0, 2, 2>), dim3(grid / 2), dim3(128), 0, 0, p, W, H, n_strips, chunk_rows, n_chunks, q, out); });
                b2 = t < b2 ? t : b2;
                t = timeit([&] { hipLaunchKernelGGL((read_strips<60, 0, 2, 4>), dim3(grid / 4), dim3(256), 0, 0, p, W, H, n_strips, chunk_rows, n_chunks, q, out); });
                b4 = t < b4 ? t : b4;
            }
            printf("frame skew %6zu float4: ... 2 / 4 adjacent strips per workgroup in lock-step: %.2f / %.2f TB/s alg\n", skew, 1.25 * gb * 59 / 60 / b2 / 1e3, 1.25 * gb * 59 / 60 / b4 / 1e3);
        }
        double best2 = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            double t = timeit([&] { hipLaunchKernelGGL((read_strips<60, 0, 2>), dim3(grid), dim3(64), 0, 0, p, W, H, n_strips, chunk_rows, n_chunks, (float4*)nullptr, out); });
            best2 = t < best2 ? t : best2;
        }
        printf("frame skew %6zu float4: read_strips alone: %.2f TB/s (59 frames)\n", skew, gb * 59 / 60 / best2 / 1e3);
    }
    return 0;
}
