// Round 5: what separates the temporal kernel's slow (37-38 us per 4K frame) from its fast (31-33 us) destination buffers?
// One process, NB level-0 candidates of three kinds (hipMalloc, 32 MB chunks, 2 MB chunks through the virtual-memory API) held at
// once, and on EACH of them (a) the library's real temporal_vec_kernel<8,4,u8> (included from csrc/; "k1" = as the library launches it,
// since round 5 with 4 waves per workgroup; "k1_x16" = its body with one wave per workgroup, blocks in runs of 16), (b) the same kernel with an
// XCD-contiguous block order, (c) kernels that only replay its address stream -- 6 byte-plane dword loads one frame ahead, four
// 1 KiB float4 store runs per frame and wave -- in several variants (block order, waves per workgroup, read-only, write-only,
// store cache policy, filler arithmetic).  Not part of the product.
//   build:  hipcc --offload-arch=gfx950 -O3 -I fovvideovdp_amd/csrc -I include tools/microbench/k1_stream.hip -o build_variants/k1_stream
//   usage:  k1_stream [n_malloc n_vmm32 n_vmm2] [pmc]      ("pmc": one warm + one launch of the real kernel and the plain replay
//                                                            per buffer, nothing else -- the target of rocprofv3 --pmc passes)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <string>
#include <vector>

#include "fvvdp_hip.h"
#include "device_common.hpp"
#include "temporal_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static const int W = 3840, H = 2160, HW = W * H, NOUT = 60, FLEN = 8, NSRC = NOUT + FLEN - 1;
// level 0 in one contiguous range (L0Addr, device_common.hpp)
static L0Addr one_range(float* o) { L0Addr a; a.lo = o; a.hi = o + (size_t)HW * 4; a.half_stride = (size_t)HW * 8; a.slot0 = 0; return a; }

// ---- the real kernel with another block order --------------------------------------------------------------------------------
// Workgroup i runs on XCD i % 8.  "XCD-contiguous": XCD x takes the x-th eighth of the pixel blocks, neighbours in dispatch order on
// one XCD are neighbours in memory.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int per = (n + 7) / 8;
    return (b % 8) * per + b / 8;
}
template <int MAP>
__global__ __launch_bounds__(64, 4) void k1_mapped(const TemporalArgs a_byval) {
    const TemporalArgs& a = *(const TemporalArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_byval;
    __shared__ float lutw[768];
    __shared__ float4 s_t[64 * 5];
    build_lutw(lutw, a.e.lut, a.C, a.w, threadIdx.x, 64);
    __syncthreads();
    const int n_blocks = (a.HW + 255) / 256;
    int block = blockIdx.x;
    if (MAP == 1) block = xcd_contiguous(block, n_blocks);
    if (MAP == 2) {                                    // XCD x takes runs of 16 consecutive blocks (64 KiB of a level-0 frame)
        const int x = block % 8, y = block / 8;
        block = ((y / 16) * 8 + x) * 16 + (y % 16);
    }
    if (block >= n_blocks) return;
    temporal_vec_cc<8, 4, SRC_U8, 1, FVVDP_EOTF_LUT>(a, lutw, s_t, block);
}

// the real kernel, WAVES adjacent pixel blocks per workgroup (one per wave; the waves share the code-value table, nothing else)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void k1_multi(const TemporalArgs a_byval) {
    const TemporalArgs& a = *(const TemporalArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_byval;
    __shared__ float lutw[768];
    __shared__ float4 s_t[WAVES][64 * 5];
    build_lutw(lutw, a.e.lut, a.C, a.w, threadIdx.x, 64 * WAVES);
    __syncthreads();
    const int n_blocks = (a.HW + 255) / 256;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int block = (int)blockIdx.x * WAVES + wave;
    if (block >= n_blocks) return;
    temporal_vec_cc<8, 4, SRC_U8, 1, FVVDP_EOTF_LUT>(a, lutw, s_t[wave], block);
}

// ---- replay of the address stream -------------------------------------------------------------------------------------------
struct ReplayArgs {
    const unsigned char* src[2];
    size_t chan_stride, frame_stride;
    int HW, n_out, n_blocks;
    float* out;
    int wg0;                 // first workgroup of this launch (a launch over a slab of the frame)
};
// MAP 0 = block order of the real kernel, 1 = XCD-contiguous, 2 = runs of 16; RD / WR = with loads / stores; WAVES per workgroup;
// dynamic LDS of 10 KB per wave holds the replay at the real kernel's 16 waves per CU (0 = as many as fit: "rp_occ8");
// AUX = cache policy of the stores (2 = nt as in the real kernel); ALU = filler multiply-adds per step and pixel
template <int MAP, bool RD, bool WR, int WAVES, int AUX, int ALU, bool STEP = false, bool TOUCH = false>
__global__ __launch_bounds__(64 * WAVES) void replay(const ReplayArgs a) {
    int wg = blockIdx.x + a.wg0;
    const int n_wg = (a.n_blocks + WAVES - 1) / WAVES;
    if (MAP == 1) wg = xcd_contiguous(wg, n_wg);
    if (MAP == 2) { const int x = wg % 8, y = wg / 8; wg = ((y / 16) * 8 + x) * 16 + (y % 16); }
    int block = wg * WAVES + (int)(threadIdx.x / 64);
    if (wg >= n_wg) return;
    const bool live = block < a.n_blocks;
    if (!live) { if (!STEP) return; block = a.n_blocks - 1; }
    const int lane = threadIdx.x % 64;
    const int p0 = block * 256;
    const int pl = min(p0 + lane * 4, a.HW - 4);
    unsigned int nx[6];
    auto fetch = [&](int f, unsigned int (&v)[6]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[s * 3 + c] = RD ? (TOUCH ? *reinterpret_cast<const unsigned int*>(a.src[s] + (size_t)f * a.frame_stride + c * a.chan_stride + pl)
                                           : __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(a.src[s] + (size_t)f * a.frame_stride + c * a.chan_stride + pl)))
                                  : (unsigned int)(f + c);
    };
    unsigned int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) soff[i] = (p0 + i * 64 + lane < a.HW) ? (unsigned int)(p0 + i * 64 + lane) * 16u : 0xFFFFFFFFu;
    const unsigned int frame_bytes = (unsigned int)a.HW * 16u;
    const int total = 7 + a.n_out;
    float hist = 0.0f;
    if (TOUCH) {
        // a read-only phase first: every wave of the launch touches all source bytes of its block (the launch is one round of resident
        // waves that start together), so that the loop below finds them in the memory-side cache and sends only its stores to the DRAM
        unsigned int acc = 0;
        for (int f = 0; f < total; ++f) {
            unsigned int tmp[6];
            fetch(f, tmp);
#pragma unroll
            for (int k = 0; k < 6; ++k) acc ^= tmp[k];
        }
        if (acc == 0x12345678u) hist = 1.0f;
    }
    fetch(0, nx);
    for (int v = 0; v < total; ++v) {
        unsigned int cur[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) cur[k] = nx[k];
        if (STEP) __builtin_amdgcn_s_barrier();          // the workgroup's waves walk the frames in step
        fetch(min(v + 1, total - 1), nx);
        float x0 = (float)(cur[0] & 0xFF) + (float)(cur[1] >> 24) + (float)(cur[2] & 0xFF00);
        float x1 = (float)(cur[3] & 0xFF) + (float)(cur[4] >> 24) + (float)(cur[5] & 0xFF00);
#pragma unroll
        for (int k = 0; k < ALU; ++k) { x0 = __builtin_fmaf(x0, 1.0001f, x1); x1 = __builtin_fmaf(x1, 0.9999f, x0); }
        hist = hist * 0.5f + x0;
        if (v < 7) continue;
        int t = v - 7;
        asm volatile("" : "+s"(t));
        if (WR && live) {
            const __amdgpu_buffer_rsrc_t o = level_rsrc(a.out + (size_t)t * a.HW * 4, frame_bytes);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{hist, x1, x0, (float)i}), o, soff[i], 0, AUX);
        } else if (hist == 1.2345e-30f) {
            a.out[0] = hist;
        }
    }
}

// The same stream in chip-wide PHASES: every wave loads the source bytes of B frames in one go and then stores B output frames; with GATE
// the loads may only be issued inside a window of the 100 MHz real-time counter that all waves see (t mod period < rwin), with GATE 2 the
// stores only outside it -- reads and writes of the whole chip separated in time (session 32: loads first, stores after = 25 us per frame,
// interleaved = 31).  One wave per workgroup, 7 history frames loaded in a prologue.
struct PhaseArgs { ReplayArgs r; unsigned int period, rwin; };
template <int B, int GATE>
__global__ __launch_bounds__(64) void replay_phased(const PhaseArgs pa) {
    const ReplayArgs& a = pa.r;
    const int block = blockIdx.x;
    if (block >= a.n_blocks) return;
    const int lane = threadIdx.x;
    const int p0 = block * 256;
    const int pl = min(p0 + lane * 4, a.HW - 4);
    unsigned int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) soff[i] = (p0 + i * 64 + lane < a.HW) ? (unsigned int)(p0 + i * 64 + lane) * 16u : 0xFFFFFFFFu;
    const unsigned int frame_bytes = (unsigned int)a.HW * 16u;
    auto fetch = [&](int f, unsigned int (&v)[6]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[s * 3 + c] = __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(a.src[s] + (size_t)f * a.frame_stride + c * a.chan_stride + pl));
    };
    auto in_window = [&]() { return ((unsigned int)__builtin_amdgcn_s_memrealtime() % pa.period) < pa.rwin; };
    float hist = 0.0f;
    {   // history
        unsigned int h[7][6];
        if (GATE >= 1) while (!in_window()) __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int k = 0; k < 7; ++k) fetch(k, h[k]);
#pragma unroll
        for (int k = 0; k < 7; ++k) hist = hist * 0.5f + (float)(h[k][0] & 0xFF) + (float)(h[k][3] >> 24) + (float)(h[k][5] & 0xFF00) + (float)(h[k][1] ^ h[k][2] ^ h[k][4]);
    }
    for (int t0 = 0; t0 < a.n_out; t0 += B) {
        unsigned int buf[B][6];
        if (GATE >= 1) while (!in_window()) __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int k = 0; k < B; ++k) fetch(min(7 + t0 + k, 6 + a.n_out), buf[k]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (GATE >= 2) while (in_window()) __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int k = 0; k < B; ++k) {
            int t = t0 + k;                      // (n_out is a multiple of B)
            const float x0 = (float)(buf[k][0] & 0xFF) + (float)(buf[k][1] >> 24) + (float)(buf[k][2] & 0xFF00);
            const float x1 = (float)(buf[k][3] & 0xFF) + (float)(buf[k][4] >> 24) + (float)(buf[k][5] & 0xFF00);
            hist = hist * 0.5f + x0;
            asm volatile("" : "+s"(t));
            const __amdgpu_buffer_rsrc_t o = level_rsrc(a.out + (size_t)t * a.HW * 4, frame_bytes);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v4f{hist, x1, x0, (float)i}), o, soff[i], 0, 2);
        }
    }
}

// plain streaming write / read of the same 7.96 GB (grid-stride, 16 B per lane): is a slow buffer slow for every pattern?
__global__ __launch_bounds__(256) void stream_write(float4* __restrict__ q, size_t n4, float v) {
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 1024;
    for (; i + 768 < n4; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v4f{v, v + u, v, v}, reinterpret_cast<v4f*>(q + i + u * 256));
    }
}
__global__ __launch_bounds__(256) void stream_read(const float4* __restrict__ p, size_t n4, float* out) {
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

// the same streaming write into TWO ranges at once (even blocks -> q0, odd blocks -> q1): do two physical chunks share whatever
// limits the write rate of one of them?
__global__ __launch_bounds__(256) void stream_write2(float4* __restrict__ q0, float4* __restrict__ q1, size_t n4_each, float v) {
    float4* q = (blockIdx.x & 1) ? q1 : q0;
    size_t i = (size_t)(blockIdx.x >> 1) * 1024 + threadIdx.x;
    const size_t stride = (size_t)(gridDim.x >> 1) * 1024;
    for (; i + 768 < n4_each; i += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v4f{v, v + u, v, v}, reinterpret_cast<v4f*>(q + i + u * 256));
    }
}

// ---- buffers ---------------------------------------------------------------------------------------------------------------
struct Buf { void* ptr; size_t size; std::vector<hipMemGenericAllocationHandle_t> h; size_t chunk; std::string kind; };
static Buf alloc_buf(size_t bytes, size_t chunk) {
    Buf b; b.chunk = chunk; b.size = bytes; b.ptr = nullptr;
    b.kind = chunk ? ("vmm" + std::to_string(chunk >> 20)) : "malloc";
    if (!chunk) { CK(hipMalloc(&b.ptr, bytes)); return b; }
    if (chunk == 1 || chunk == 2) {          // other memory kinds of the runtime: fine-grained (coherent) / uncached device memory
        b.kind = chunk == 1 ? "fine" : "uncached";
        if (hipExtMallocWithFlags(&b.ptr, bytes, chunk == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); b.ptr = nullptr; }
        return b;
    }
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

__global__ void fill_u8(unsigned char* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (unsigned char)((i * 2654435761u) >> 13);
}
__global__ void checksum_f4(const float4* p, size_t n4, double* out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        s += (double)v.x + 2.0 * v.y + 3.0 * v.z + 5.0 * v.w;
    }
    atomicAdd(out, s);
}

static size_t out_bytes_c(int hw) { return (size_t)NOUT * hw * 16; }

int main(int argc, char** argv) {
    int n_malloc = 4, n_v32 = 4, n_v2 = 2, n_other = 0;
    bool pmc = false, regions = false, zones = false, balanced = false, spread = false; bool decoupled = false;
    std::vector<int> nums;
    for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "pmc")) pmc = true; else if (!strcmp(argv[i], "regions")) regions = true; else if (!strcmp(argv[i], "zones")) zones = true; else if (!strcmp(argv[i], "balanced")) balanced = true; else if (!strcmp(argv[i], "spread")) spread = true; else if (!strcmp(argv[i], "decoupled")) decoupled = true; else nums.push_back(atoi(argv[i])); }
    if (nums.size() >= 3) { n_malloc = nums[0]; n_v32 = nums[1]; n_v2 = nums[2]; }
    if (nums.size() >= 4) n_other = nums[3];
    if (zones) {
        // G physical chunks of 1 GiB in allocation order, each mapped on its own; write rate of every chunk alone, of chunk 0
        // together with chunk i, and of neighbours (i, i+1): which chunks share the resource that caps one chunk at ~5.3 TB/s?
        const int G = nums.size() >= 1 ? nums[0] : 160;
        const size_t gib = (size_t)1 << 30, n4 = gib / 16;
        int dev = 0; CK(hipGetDevice(&dev));
        hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc)); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        std::vector<float4*> ch;
        for (int g = 0; g < G; ++g) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, gib, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            void* va = nullptr;
            CK(hipMemAddressReserve(&va, gib, (size_t)2 << 20, nullptr, 0));
            CK(hipMemMap(va, gib, 0, h, 0));
            CK(hipMemSetAccess(va, gib, &acc, 1));
            ch.push_back((float4*)va);
        }
        const int n = (int)ch.size();
        printf("%d chunks of 1 GiB (allocation order)\n", n);
        for (int g = 0; g < n; ++g) CK(hipMemsetAsync(ch[g], 0, gib, 0));
        CK(hipDeviceSynchronize());
        auto one = [&](int i) { auto f = [&] { hipLaunchKernelGGL(stream_write, dim3(8192), dim3(256), 0, 0, ch[i], n4, 1.0f); }; f(); return gib / time_us(f, 3) / 1e6; };
        auto two = [&](int i, int j) { auto f = [&] { hipLaunchKernelGGL(stream_write2, dim3(16384), dim3(256), 0, 0, ch[i], ch[j], n4, 1.0f); }; f(); return 2.0 * gib / time_us(f, 3) / 1e6; };
        printf("%-5s %-8s %-10s %-12s %-14s\n", "chunk", "alone", "with #0", "with next", "with #n/2");
        for (int i = 0; i < n; ++i)
            printf("%-5d %-8.2f %-10.2f %-12.2f %-14.2f\n", i, one(i), i ? two(0, i) : 0.0, i + 1 < n ? two(i, i + 1) : 0.0, two(i, n / 2 == i ? 0 : n / 2));
        // and a few chunks in full pairwise detail
        const int pick[8] = {0, n / 8, n / 4, 3 * n / 8, n / 2, 5 * n / 8, 3 * n / 4, n - 1};
        printf("pairwise write rate [TB/s] of chunks");
        for (int a = 0; a < 8; ++a) printf(" %d", pick[a]);
        printf("\n");
        for (int a = 0; a < 8; ++a) {
            for (int b = 0; b < 8; ++b) printf(" %5.2f", a == b ? one(pick[a]) : two(pick[a], pick[b]));
            printf("\n");
        }
        return 0;
    }
    const size_t src_bytes = (size_t)NSRC * 3 * HW;
    if (spread) {
        // Level-0 candidates whose two halves come from two allocation points S GiB apart (a spacer of S physical GiB is allocated in
        // between and released afterwards), the halves interleaved chunk by chunk.  usage: k1_stream <chunk MB> spread
        const size_t chunk = (size_t)(nums.size() >= 1 ? nums[0] : 32) << 20, gib = (size_t)1 << 30;
        int dev = 0; CK(hipGetDevice(&dev));
        hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc)); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        unsigned char* srcb[2];
        for (int s2 = 0; s2 < 2; ++s2) { CK(hipMalloc((void**)&srcb[s2], src_bytes)); hipLaunchKernelGGL(fill_u8, dim3(4096), dim3(256), 0, 0, srcb[s2], src_bytes); }
        float* d_lut2; CK(hipMalloc((void**)&d_lut2, 256 * sizeof(float)));
        { float h[256]; for (int i = 0; i < 256; ++i) h[i] = 0.6f + 199.4f * powf(i / 255.0f, 2.2f); CK(hipMemcpy(d_lut2, h, sizeof(h), hipMemcpyHostToDevice)); }
        int* d_oob2; CK(hipMalloc((void**)&d_oob2, 64)); CK(hipMemset(d_oob2, 0, 64));
        const size_t n_chunks = (out_bytes_c(HW) + chunk - 1) / chunk;
        void* va = nullptr; CK(hipMemAddressReserve(&va, n_chunks * chunk, chunk, nullptr, 0));
        auto wall = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
        printf("chunks of %zu MB, %zu per level 0\n%-28s %8s %8s %8s %10s\n", chunk >> 20, n_chunks, "layout", "write", "rp_w", "k1", "build ms");
        const int spacers[] = {0, 4, 8, 16, 24, 32, 48, 64, 96, 0};
        for (int sp : spacers) {
            const double t0 = wall();
            std::vector<hipMemGenericAllocationHandle_t> h1, h2, hs;
            bool okc = true;
            auto grab = [&](std::vector<hipMemGenericAllocationHandle_t>& v, size_t n, size_t sz) {
                for (size_t k = 0; k < n && okc; ++k) { hipMemGenericAllocationHandle_t h; okc = hipMemCreate(&h, sz, &prop, 0) == hipSuccess; if (okc) v.push_back(h); }
            };
            grab(h1, (n_chunks + 1) / 2, chunk);
            grab(hs, (size_t)sp, gib);
            grab(h2, n_chunks / 2, chunk);
            for (auto h : hs) (void)hipMemRelease(h);
            if (!okc) { (void)hipGetLastError(); printf("spacer %d GiB: allocation failed\n", sp); for (auto h : h1) (void)hipMemRelease(h); for (auto h : h2) (void)hipMemRelease(h); continue; }
            for (size_t k = 0; k < n_chunks; ++k) CK(hipMemMap((char*)va + k * chunk, chunk, 0, (k & 1) ? h2[k / 2] : h1[k / 2], 0));
            CK(hipMemSetAccess(va, n_chunks * chunk, &acc, 1));
            const double t_build = wall() - t0;
            const size_t n4 = (size_t)NOUT * HW;
            auto fw = [&] { hipLaunchKernelGGL(stream_write, dim3(32768), dim3(256), 0, 0, (float4*)va, n4, 1.0f); };
            TemporalArgs ta2; memset(&ta2, 0, sizeof(ta2));
            ta2.src[0] = srcb[0]; ta2.src[1] = srcb[1]; ta2.chan_stride = HW; ta2.frame_stride = (size_t)3 * HW; ta2.C = 3; ta2.HW = HW;
            ta2.e.kind = FVVDP_EOTF_LUT; ta2.e.lut = d_lut2; ta2.w[0] = 0.2126f; ta2.w[1] = 0.7152f; ta2.w[2] = 0.0722f;
            ta2.n_out = NOUT; ta2.fl = FLEN; ta2.oob = d_oob2; ta2.out = one_range((float*)va);
            for (int q = 0; q < FLEN; ++q) { ta2.taps2[q][0] = 0.3f / (1 + q); ta2.taps2[q][1] = (q & 1) ? -0.1f : 0.1f; }
            for (int u = 0; u < NSRC; ++u) ta2.idx[u] = ta2.idx1[u] = u;
            ReplayArgs ra2; ra2.wg0 = 0; ra2.src[0] = srcb[0]; ra2.src[1] = srcb[1]; ra2.chan_stride = HW; ra2.frame_stride = (size_t)3 * HW; ra2.HW = HW; ra2.n_out = NOUT;
            ra2.n_blocks = (HW + 255) / 256; ra2.out = (float*)va;
            auto fk = [&] { hipLaunchKernelGGL((temporal_vec_kernel<8, 4, SRC_U8, 1>), dim3((ra2.n_blocks + k1_wpb(8) - 1) / k1_wpb(8)), dim3(64 * k1_wpb(8)), 0, 0, ta2); };
            auto fp = [&] { hipLaunchKernelGGL((replay<0, false, true, 1, 2, 0>), dim3(ra2.n_blocks), dim3(64), 10240, 0, ra2); };
            fw(); fk(); fp(); CK(hipDeviceSynchronize());
            const double tw = time_us(fw, 5), tp = time_us(fp, 5), tk = time_us(fk, 5);
            char name[64]; snprintf(name, sizeof(name), "halves %d GiB apart", sp);
            printf("%-28s %8.2f %8.2f %8.2f %10.1f\n", name, n4 * 16.0 / tw / 1e6, tp / NOUT, tk / NOUT, t_build);
            fflush(stdout);
            CK(hipMemUnmap(va, n_chunks * chunk));
            for (auto h : h1) (void)hipMemRelease(h);
            for (auto h : h2) (void)hipMemRelease(h);
        }
        return 0;
    }
    if (balanced) {
        // Level-0 candidates assembled from 32 MB physical chunks of KNOWN class (groups of 32 chunks = 1 GiB classified by the
        // pair-write rate against the first group): one class only, both classes interleaved at several granularities and ratios.
        // On each: streaming write / read rate, the store-stream replay and the real temporal kernel.
        // usage: k1_stream <max groups> <chunk MB> balanced
        const size_t chunk = (size_t)(nums.size() >= 2 ? nums[1] : 32) << 20, gib = (size_t)1 << 30, n4g = gib / 16;
        const int per = (int)(gib / chunk), max_groups = nums.size() >= 1 ? nums[0] : 120;
        int dev = 0; CK(hipGetDevice(&dev));
        hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
        hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc)); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        unsigned char* srcb[2];
        for (int s2 = 0; s2 < 2; ++s2) { CK(hipMalloc((void**)&srcb[s2], src_bytes)); hipLaunchKernelGGL(fill_u8, dim3(4096), dim3(256), 0, 0, srcb[s2], src_bytes); }
        float* d_lut2; CK(hipMalloc((void**)&d_lut2, 256 * sizeof(float)));
        { float h[256]; for (int i = 0; i < 256; ++i) h[i] = 0.6f + 199.4f * powf(i / 255.0f, 2.2f); CK(hipMemcpy(d_lut2, h, sizeof(h), hipMemcpyHostToDevice)); }
        int* d_oob2; CK(hipMalloc((void**)&d_oob2, 64)); CK(hipMemset(d_oob2, 0, 64));
        void* probe_va = nullptr; CK(hipMemAddressReserve(&probe_va, 2 * gib, chunk, nullptr, 0));
        std::vector<std::vector<hipMemGenericAllocationHandle_t>> groups;
        std::vector<int> cls;
        auto map_group = [&](int g, char* va) { for (int k = 0; k < per; ++k) CK(hipMemMap(va + k * chunk, chunk, 0, groups[g][k], 0)); CK(hipMemSetAccess(va, gib, &acc, 1)); };
        auto unmap_group = [&](char* va) { CK(hipMemUnmap(va, gib)); };
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto wall = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
        const double t_start = wall();
        double solo = 0.0;
        int nA = 0, nB = 0;
        for (int g = 0; g < max_groups && (nA < 12 || nB < 12); ++g) {
            std::vector<hipMemGenericAllocationHandle_t> hs;
            bool okc = true;
            for (int k = 0; k < per && okc; ++k) { hipMemGenericAllocationHandle_t h; okc = hipMemCreate(&h, chunk, &prop, 0) == hipSuccess; if (okc) hs.push_back(h); }
            if (!okc) { (void)hipGetLastError(); for (auto h : hs) (void)hipMemRelease(h); break; }
            groups.push_back(hs);
            if (g == 0) {
                map_group(0, (char*)probe_va);
                auto f = [&] { hipLaunchKernelGGL(stream_write, dim3(8192), dim3(256), 0, 0, (float4*)probe_va, n4g, 1.0f); };
                f(); solo = gib / time_us(f, 3) / 1e6;
                cls.push_back(0); ++nA;
                continue;
            }
            map_group(g, (char*)probe_va + gib);
            auto f = [&] { hipLaunchKernelGGL(stream_write2, dim3(16384), dim3(256), 0, 0, (float4*)probe_va, (float4*)((char*)probe_va + gib), n4g, 1.0f); };
            f(); const double pair = 2.0 * gib / time_us(f, 3) / 1e6;
            unmap_group((char*)probe_va + gib);
            const int c2 = pair > 1.15 * solo ? 1 : 0;
            cls.push_back(c2); if (c2) ++nB; else ++nA;
        }
        unmap_group((char*)probe_va);
        const double t_class = wall() - t_start;
        printf("chunks of %zu MB: classified %zu groups of 1 GiB in %.1f ms (solo %.2f TB/s): ", chunk >> 20, groups.size(), t_class, solo);
        for (int c2 : cls) printf("%c", c2 ? 'B' : 'A');
        printf("\n");
        std::vector<hipMemGenericAllocationHandle_t> hA, hB;
        for (size_t g = 0; g < groups.size(); ++g) for (auto h : groups[g]) (cls[g] ? hB : hA).push_back(h);
        const size_t n_chunks = (out_bytes_c(HW) + chunk - 1) / chunk;
        void* va = nullptr; CK(hipMemAddressReserve(&va, n_chunks * chunk, chunk, nullptr, 0));
        struct Cfg { const char* name; int runA, runB; };      // runA chunks of class A, then runB of class B, repeating
        const int u = per;                                     // chunks per GiB
        const Cfg cfgs[] = {{"A only", 1, 0}, {"B only", 0, 1}, {"A/B 1 chunk", 1, 1}, {"A/B 2 chunks", 2, 2}, {"A/B 1 GiB", u, u}, {"A/B 2 GiB", 2 * u, 2 * u},
                            {"A/B 4 GiB", 4 * u, 4 * u}, {"3A/1B chunks", 3, 1}, {"1A/3B chunks", 1, 3}, {"7A/1B chunks", 7, 1}, {"6 GiB A, 2 GiB B", 6 * u, 2 * u},
                            {"A/B 1 chunk again", 1, 1}};
        printf("%-18s %10s %10s %10s %10s   (write / read TB/s of the whole 7.96 GB; store-stream replay and real temporal kernel us per 4K frame)\n", "level-0 layout", "write", "read", "rp_w", "k1");
        for (const Cfg& cf : cfgs) {
            size_t ia = 0, ib = 0, k = 0; bool okm = true;
            while (k < n_chunks && okm) {
                for (int r = 0; r < cf.runA && k < n_chunks && okm; ++r) { okm = ia < hA.size(); if (okm) CK(hipMemMap((char*)va + k++ * chunk, chunk, 0, hA[ia++], 0)); }
                for (int r = 0; r < cf.runB && k < n_chunks && okm; ++r) { okm = ib < hB.size(); if (okm) CK(hipMemMap((char*)va + k++ * chunk, chunk, 0, hB[ib++], 0)); }
            }
            if (!okm) { printf("%-18s not enough chunks of one class\n", cf.name); if (k) CK(hipMemUnmap(va, k * chunk)); continue; }
            CK(hipMemSetAccess(va, n_chunks * chunk, &acc, 1));
            const size_t n4 = (size_t)NOUT * HW;
            auto fw = [&] { hipLaunchKernelGGL(stream_write, dim3(32768), dim3(256), 0, 0, (float4*)va, n4, 1.0f); };
            auto fr = [&] { hipLaunchKernelGGL(stream_read, dim3(32768), dim3(256), 0, 0, (const float4*)va, n4, (float*)d_oob2); };
            TemporalArgs ta2; memset(&ta2, 0, sizeof(ta2));
            ta2.src[0] = srcb[0]; ta2.src[1] = srcb[1]; ta2.chan_stride = HW; ta2.frame_stride = (size_t)3 * HW; ta2.C = 3; ta2.HW = HW;
            ta2.e.kind = FVVDP_EOTF_LUT; ta2.e.lut = d_lut2; ta2.w[0] = 0.2126f; ta2.w[1] = 0.7152f; ta2.w[2] = 0.0722f;
            ta2.n_out = NOUT; ta2.fl = FLEN; ta2.oob = d_oob2; ta2.out = one_range((float*)va);
            for (int q = 0; q < FLEN; ++q) { ta2.taps2[q][0] = 0.3f / (1 + q); ta2.taps2[q][1] = (q & 1) ? -0.1f : 0.1f; }
            for (int u = 0; u < NSRC; ++u) ta2.idx[u] = ta2.idx1[u] = u;
            ReplayArgs ra2; ra2.wg0 = 0; ra2.src[0] = srcb[0]; ra2.src[1] = srcb[1]; ra2.chan_stride = HW; ra2.frame_stride = (size_t)3 * HW; ra2.HW = HW; ra2.n_out = NOUT;
            ra2.n_blocks = (HW + 255) / 256; ra2.out = (float*)va;
            auto fk = [&] { hipLaunchKernelGGL((temporal_vec_kernel<8, 4, SRC_U8, 1>), dim3((ra2.n_blocks + k1_wpb(8) - 1) / k1_wpb(8)), dim3(64 * k1_wpb(8)), 0, 0, ta2); };
            auto fp = [&] { hipLaunchKernelGGL((replay<0, false, true, 1, 2, 0>), dim3(ra2.n_blocks), dim3(64), 10240, 0, ra2); };
            fw(); fr(); fk(); fp(); CK(hipDeviceSynchronize());
            const double tw = time_us(fw, 5), tr = time_us(fr, 5), tp = time_us(fp, 5), tk = time_us(fk, 5);
            printf("%-18s %10.2f %10.2f %10.2f %10.2f\n", cf.name, n4 * 16.0 / tw / 1e6, n4 * 16.0 / tr / 1e6, tp / NOUT, tk / NOUT);
            fflush(stdout);
            CK(hipMemUnmap(va, n_chunks * chunk));
        }
        return 0;
    }
    unsigned char* src[2];
    for (int s = 0; s < 2; ++s) { CK(hipMalloc((void**)&src[s], src_bytes)); hipLaunchKernelGGL(fill_u8, dim3(4096), dim3(256), 0, 0, src[s], src_bytes); }
    float* d_lut; CK(hipMalloc((void**)&d_lut, 256 * sizeof(float)));
    { float h[256]; for (int i = 0; i < 256; ++i) h[i] = 0.6f + 199.4f * powf(i / 255.0f, 2.2f); CK(hipMemcpy(d_lut, h, sizeof(h), hipMemcpyHostToDevice)); }
    int* d_oob; CK(hipMalloc((void**)&d_oob, 64)); CK(hipMemset(d_oob, 0, 64));
    double* d_sum; CK(hipMalloc((void**)&d_sum, 8));
    const size_t out_bytes = (size_t)NOUT * HW * 16;
    std::vector<Buf> bufs;
    // interleave the kinds so that allocation order does not line up with kind
    for (int i = 0; i < std::max(std::max(n_malloc, n_other), std::max(n_v32, n_v2)); ++i) {
        if (i < n_malloc) bufs.push_back(alloc_buf(out_bytes, 0));
        if (i < n_v32) bufs.push_back(alloc_buf(out_bytes, (size_t)32 << 20));
        if (i < n_v2) bufs.push_back(alloc_buf(out_bytes, (size_t)2 << 20));
        if (i < n_other) { Buf f = alloc_buf(out_bytes, 1); if (f.ptr) bufs.push_back(f); Buf u = alloc_buf(out_bytes, 2); if (u.ptr) bufs.push_back(u); }
    }
    for (auto& b : bufs) CK(hipMemset(b.ptr, 0, out_bytes));
    CK(hipDeviceSynchronize());
    TemporalArgs ta; memset(&ta, 0, sizeof(ta));
    ta.src[0] = src[0]; ta.src[1] = src[1]; ta.chan_stride = HW; ta.frame_stride = (size_t)3 * HW; ta.C = 3; ta.HW = HW;
    ta.e.kind = FVVDP_EOTF_LUT; ta.e.lut = d_lut; ta.w[0] = 0.2126f; ta.w[1] = 0.7152f; ta.w[2] = 0.0722f;
    ta.n_out = NOUT; ta.fl = FLEN; ta.oob = d_oob; ta.ticket = nullptr;
    for (int k = 0; k < FLEN; ++k) { ta.taps2[k][0] = 0.3f / (1 + k); ta.taps2[k][1] = (k & 1) ? -0.1f : 0.1f; }
    for (int u = 0; u < NSRC; ++u) ta.idx[u] = ta.idx1[u] = u;
    ReplayArgs ra; ra.wg0 = 0; ra.src[0] = src[0]; ra.src[1] = src[1]; ra.chan_stride = HW; ra.frame_stride = (size_t)3 * HW; ra.HW = HW; ra.n_out = NOUT;
    ra.n_blocks = (HW + 255) / 256;
    const int nb = ra.n_blocks, nb8 = (nb + 7) / 8 * 8;
    struct Variant { const char* name; std::function<void(float*)> run; };
    std::vector<Variant> vars;
    vars.push_back({"k1", [&](float* o) { TemporalArgs a = ta; a.out = one_range(o); hipLaunchKernelGGL((temporal_vec_kernel<8, 4, SRC_U8, 1>), dim3((nb + k1_wpb(8) - 1) / k1_wpb(8)), dim3(64 * k1_wpb(8)), 0, 0, a); }});
    vars.push_back({"rp", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(nb), dim3(64), 10240, 0, a); }});
    if (!pmc) {
        vars.push_back({"k1_xcd", [&](float* o) { TemporalArgs a = ta; a.out = one_range(o); hipLaunchKernelGGL((k1_mapped<1>), dim3(nb8), dim3(64), 0, 0, a); }});
        vars.push_back({"k1_x16", [&](float* o) { TemporalArgs a = ta; a.out = one_range(o); hipLaunchKernelGGL((k1_mapped<2>), dim3((nb + 127) / 128 * 128), dim3(64), 0, 0, a); }});
        vars.push_back({"rp_alu", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 1, 2, 48>), dim3(nb), dim3(64), 10240, 0, a); }});
        vars.push_back({"rp_xcd", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<1, true, true, 1, 2, 0>), dim3(nb8), dim3(64), 10240, 0, a); }});
        vars.push_back({"rp_x16", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<2, true, true, 1, 2, 0>), dim3((nb + 127) / 128 * 128), dim3(64), 10240, 0, a); }});
        vars.push_back({"rp_w", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, false, true, 1, 2, 0>), dim3(nb), dim3(64), 10240, 0, a); }});
        vars.push_back({"rp_r", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, false, 1, 2, 0>), dim3(nb), dim3(64), 10240, 0, a); }});
        vars.push_back({"rp_4w", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 4, 2, 0>), dim3((nb + 3) / 4), dim3(256), 40960, 0, a); }});
        vars.push_back({"rp_4wx", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<1, true, true, 4, 2, 0>), dim3(((nb + 3) / 4 + 7) / 8 * 8), dim3(256), 40960, 0, a); }});
        vars.push_back({"rp_occ8", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(nb), dim3(64), 0, 0, a); }});
        vars.push_back({"k1_2w", [&](float* o) { TemporalArgs a = ta; a.out = one_range(o); hipLaunchKernelGGL((k1_multi<2>), dim3((nb + 1) / 2), dim3(128), 0, 0, a); }});
        vars.push_back({"k1_4w", [&](float* o) { TemporalArgs a = ta; a.out = one_range(o); hipLaunchKernelGGL((k1_multi<4>), dim3((nb + 3) / 4), dim3(256), 0, 0, a); }});
        vars.push_back({"k1_8w", [&](float* o) { TemporalArgs a = ta; a.out = one_range(o); hipLaunchKernelGGL((k1_multi<8>), dim3((nb + 7) / 8), dim3(512), 0, 0, a); }});
        vars.push_back({"rp_2w", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 2, 2, 0>), dim3((nb + 1) / 2), dim3(128), 20480, 0, a); }});
        vars.push_back({"rp_8w", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 8, 2, 0>), dim3((nb + 7) / 8), dim3(512), 65536, 0, a); }});
        vars.push_back({"rp_16w", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 16, 2, 0>), dim3((nb + 15) / 16), dim3(1024), 65536, 0, a); }});
        vars.push_back({"rp_4ws", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 4, 2, 0, true>), dim3((nb + 3) / 4), dim3(256), 40960, 0, a); }});
        vars.push_back({"rp_8ws", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 8, 2, 0, true>), dim3((nb + 7) / 8), dim3(512), 65536, 0, a); }});
        vars.push_back({"rp_16ws", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 16, 2, 0, true>), dim3((nb + 15) / 16), dim3(1024), 65536, 0, a); }});
        vars.push_back({"rp_occ2", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(nb), dim3(64), 20480, 0, a); }});
        vars.push_back({"rp_occ1", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(nb), dim3(64), 40960, 0, a); }});
        // the frame in 8 slabs, one launch each (all waves of a launch start together and stay close to in step)
        vars.push_back({"rp_slab8", [&](float* o) { for (int s8 = 0; s8 < 8; ++s8) { ReplayArgs a = ra; a.out = o; a.wg0 = s8 * (nb8 / 8);
                                                     hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(std::min(nb8 / 8, nb - a.wg0)), dim3(64), 10240, 0, a); } }});
        vars.push_back({"rp_slab4", [&](float* o) { for (int s4 = 0; s4 < 4; ++s4) { ReplayArgs a = ra; a.out = o; a.wg0 = s4 * (nb8 / 4);
                                                     hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(std::min(nb8 / 4, nb - a.wg0)), dim3(64), 10240, 0, a); } }});
        // one round of resident waves only (4096 blocks): us per frame scaled to the whole frame (x n_blocks / 4096)
        vars.push_back({"rp_1st*", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(4096), dim3(64), 10240, 0, a); }});
        vars.push_back({"wr_seq", [&](float* o) { hipLaunchKernelGGL(stream_write, dim3(32768), dim3(256), 0, 0, (float4*)o, (size_t)NOUT * HW, 1.0f); }});
        vars.push_back({"rd_seq", [&](float* o) { hipLaunchKernelGGL(stream_read, dim3(32768), dim3(256), 0, 0, (const float4*)o, (size_t)NOUT * HW, (float*)d_sum); }});
        vars.push_back({"rp_st0", [&](float* o) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 1, 0, 0>), dim3(nb), dim3(64), 10240, 0, a); }});
    }
    if (regions) {
        // streaming-write rate of every GiB of every buffer (TB/s; 1 GiB is four times the memory-side cache), then of the whole buffer
        const size_t gib4 = ((size_t)1 << 30) / 16;
        printf("%-3s %-8s %-16s  write rate of GiB 0, 1, ... [TB/s] | whole buffer | real kernel us/frame\n", "buf", "kind", "address");
        for (int rnd = 0; rnd < 2; ++rnd)
        for (size_t b = 0; b < bufs.size(); ++b) {
            printf("%-3zu %-8s %-16p ", b, bufs[b].kind.c_str(), bufs[b].ptr);
            const size_t n4 = (size_t)NOUT * HW;
            for (size_t g0 = 0; g0 + gib4 / 2 < n4; g0 += gib4) {
                float4* q = (float4*)bufs[b].ptr + g0;
                const size_t len = std::min(gib4, n4 - g0);
                auto f = [&] { hipLaunchKernelGGL(stream_write, dim3(8192), dim3(256), 0, 0, q, len, 1.0f); };
                f(); CK(hipDeviceSynchronize());
                const double us = time_us(f, 5);
                printf(" %5.2f", len * 16.0 / us / 1e6);
            }
            auto fw = [&] { hipLaunchKernelGGL(stream_write, dim3(32768), dim3(256), 0, 0, (float4*)bufs[b].ptr, n4, 1.0f); };
            fw(); CK(hipDeviceSynchronize());
            printf(" | %5.2f", n4 * 16.0 / time_us(fw, 5) / 1e6);
            auto fk = [&] { TemporalArgs a = ta; a.out = one_range((float*)bufs[b].ptr); hipLaunchKernelGGL((temporal_vec_kernel<8, 4, SRC_U8, 1>), dim3((nb + k1_wpb(8) - 1) / k1_wpb(8)), dim3(64 * k1_wpb(8)), 0, 0, a); };
            fk(); CK(hipDeviceSynchronize());
            printf(" | %5.2f\n", time_us(fk, 5) / NOUT);
            fflush(stdout);
        }
        return 0;
    }
    if (decoupled) {
        // Loads and stores of the temporal kernel's stream coupled in one wave (rp), each alone (rp_r, rp_w), and DECOUPLED: the load-only
        // and the store-only replay at the same time on two streams (5 KB of LDS each: 16 + 16 waves per CU) -- same addresses, same
        // instruction shapes, but no wave waits for its own loads before it stores.  Last: plain streams in the same proportions.
        hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
        hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        auto both = [&](std::function<void(hipStream_t)> fa, std::function<void(hipStream_t)> fb) {
            std::vector<double> v;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, s0)); CK(hipStreamWaitEvent(s1, e0, 0));
                fa(s0); fb(s1);
                CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s0, e1, 0)); CK(hipEventRecord(e2, s0)); CK(hipEventSynchronize(e2));
                float ms; CK(hipEventElapsedTime(&ms, e0, e2));
                if (rep) v.push_back(ms * 1e3);
            }
            std::sort(v.begin(), v.end());
            return v[v.size() / 2] / NOUT;
        };
        float* sink = nullptr; CK(hipMalloc(&sink, 64));
        const size_t src4 = (size_t)NSRC * 3 * HW / 16, out4 = (size_t)NOUT * HW;
        printf("%-3s %-8s %10s %10s %10s %12s %12s %14s   (us per 4K frame)\n", "buf", "kind", "rp", "rp_r", "rp_w", "rp_r || rp_w", "rp_4w", "plain r || w");
        for (size_t b = 0; b < bufs.size(); ++b) {
            float* o = (float*)bufs[b].ptr;
            auto one = [&](std::function<void(hipStream_t)> f) { return both(f, [](hipStream_t) {}); };
            const double t_rp = one([&](hipStream_t st) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(nb), dim3(64), 10240, st, a); });
            const double t_r = one([&](hipStream_t st) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, false, 1, 2, 0>), dim3(nb), dim3(64), 10240, st, a); });
            const double t_w = one([&](hipStream_t st) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, false, true, 1, 2, 0>), dim3(nb), dim3(64), 10240, st, a); });
            const double t_rw = both([&](hipStream_t st) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, false, 1, 2, 0>), dim3(nb), dim3(64), 5120, st, a); },
                                     [&](hipStream_t st) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, false, true, 1, 2, 0>), dim3(nb), dim3(64), 5120, st, a); });
            const double t_4w = one([&](hipStream_t st) { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, true, 4, 2, 0>), dim3((nb + 3) / 4), dim3(256), 40960, st, a); });
            const double t_pl = both([&](hipStream_t st) { hipLaunchKernelGGL(stream_read, dim3(8192), dim3(256), 0, st, (const float4*)src[0], src4, sink);
                                                           hipLaunchKernelGGL(stream_read, dim3(8192), dim3(256), 0, st, (const float4*)src[1], src4, sink); },
                                     [&](hipStream_t st) { hipLaunchKernelGGL(stream_write, dim3(24576), dim3(256), 0, st, (float4*)o, out4, 1.0f); });
            printf("%-3zu %-8s %10.2f %10.2f %10.2f %12.2f %12.2f %14.2f", b, bufs[b].kind.c_str(), t_rp, t_r, t_w, t_rw, t_4w, t_pl);
            // the load-only replay throttled by its LDS footprint (fewer waves per CU): do the loads have to END early for the gain, or is it
            // enough that no wave waits for its own loads?  per footprint: when the load kernel ended / when both had ended (us per frame)
            static hipEvent_t er = nullptr; if (!er) CK(hipEventCreate(&er));
            for (int kb : {20, 40, 64, 100}) {
                std::vector<double> vt, vr;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, s0)); CK(hipStreamWaitEvent(s1, e0, 0));
                    { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, false, true, 1, 2, 0>), dim3(nb), dim3(64), 2048, s1, a); }
                    { ReplayArgs a = ra; a.out = o; hipLaunchKernelGGL((replay<0, true, false, 1, 2, 0>), dim3(nb), dim3(64), kb * 1024, s0, a); }
                    CK(hipEventRecord(er, s0));
                    CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s0, e1, 0)); CK(hipEventRecord(e2, s0)); CK(hipEventSynchronize(e2));
                    float ms, mr; CK(hipEventElapsedTime(&ms, e0, e2)); CK(hipEventElapsedTime(&mr, e0, er));
                    if (rep) { vt.push_back(ms * 1e3 / NOUT); vr.push_back(mr * 1e3 / NOUT); }
                }
                std::sort(vt.begin(), vt.end()); std::sort(vr.begin(), vr.end());
                printf("  | %dK: %.1f/%.1f", kb, vr[1], vt[1]);
            }
            printf("\n");
            // the frame in N launches (one round of resident waves each or less), without / with a read-only phase at the start of every launch
            printf("      slabs, plain / touch-first:");
            for (int ns : {8, 16, 32, 64}) {
                const int per = (nb + ns - 1) / ns;
                const double t0 = one([&](hipStream_t st) { for (int k = 0; k < ns; ++k) { ReplayArgs a = ra; a.out = o; a.wg0 = k * per; if (a.wg0 >= nb) break;
                    hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0>), dim3(std::min(per, nb - a.wg0)), dim3(64), 10240, st, a); } });
                const double t1 = one([&](hipStream_t st) { for (int k = 0; k < ns; ++k) { ReplayArgs a = ra; a.out = o; a.wg0 = k * per; if (a.wg0 >= nb) break;
                    hipLaunchKernelGGL((replay<0, true, true, 1, 2, 0, false, true>), dim3(std::min(per, nb - a.wg0)), dim3(64), 10240, st, a); } });
                printf("  %d: %.2f / %.2f", ns, t0, t1);
            }
            printf("\n");
            // chip-wide phases (replay_phased): B = 10 frames per wave and period, loads gated into a window of the real-time counter
            printf("      phased B=10: ungated %.2f", one([&](hipStream_t st) { PhaseArgs pa; pa.r = ra; pa.r.out = o; pa.period = 3000; pa.rwin = 750;
                                                              hipLaunchKernelGGL((replay_phased<10, 0>), dim3(nb), dim3(64), 10240, st, pa); }));
            for (unsigned int per : {2800u, 3100u, 3400u, 3800u})
                for (unsigned int duty : {20u, 28u}) {
                    auto go = [&](int gate) { return one([&](hipStream_t st) { PhaseArgs pa; pa.r = ra; pa.r.out = o; pa.period = per; pa.rwin = per * duty / 100;
                        if (gate == 1) hipLaunchKernelGGL((replay_phased<10, 1>), dim3(nb), dim3(64), 10240, st, pa);
                        else hipLaunchKernelGGL((replay_phased<10, 2>), dim3(nb), dim3(64), 10240, st, pa); }); };
                    printf(" | %u/%u%%: %.2f %.2f", per, duty, go(1), go(2));
                }
            printf("   (period in 10 ns ticks / share of it open for loads: loads gated, loads and stores gated)\n");
            fflush(stdout);
        }
        return 0;
    }
    printf("%-3s %-8s %-16s", "buf", "kind", "address");
    for (auto& v : vars) printf(" %8s", v.name);
    printf("   (us per 4K frame, median of %d launches of 60 frames)\n", pmc ? 1 : 5);
    std::vector<std::vector<double>> tab(bufs.size());
    for (int rnd = 0; rnd < (pmc ? 1 : 2); ++rnd) {
        for (size_t b = 0; b < bufs.size(); ++b) {
            tab[b].clear();
            printf("%-3zu %-8s %-16p", b, bufs[b].kind.c_str(), bufs[b].ptr);
            for (auto& v : vars) {
                float* o = (float*)bufs[b].ptr;
                v.run(o); CK(hipDeviceSynchronize());
                double us = time_us([&] { v.run(o); }, pmc ? 1 : 5);
                if (strchr(v.name, '*')) us *= (double)nb / 4096.0;        // one round of resident waves, scaled to the frame
                tab[b].push_back(us / NOUT);
                printf(" %8.2f", us / NOUT);
            }
            printf("\n"); fflush(stdout);
        }
        if (rnd == 0 && !pmc) printf("-- second round, same buffers\n");
    }
    if (!pmc) {
        // Which buffers differ in class?  Streaming write into the first half of buffer i and the first half of buffer j at once
        // (3.98 GB each) [TB/s]; the diagonal is the whole buffer i.  Two slow buffers that reach 7 TB/s together are of different classes:
        // a level 0 with its even frames in one and its odd frames in the other would be the fast mode.
        const size_t half4 = (size_t)NOUT * HW / 2;
        printf("pairwise streaming-write rate [TB/s], first halves of buffers i and j together (diagonal: buffer i alone, whole)\n     ");
        for (size_t j = 0; j < bufs.size(); ++j) printf(" %5zu", j);
        printf("\n");
        for (size_t i = 0; i < bufs.size(); ++i) {
            printf("%-2zu %-3.3s", i, bufs[i].kind.c_str());
            for (size_t j = 0; j < bufs.size(); ++j) {
                auto f = [&] {
                    if (i == j) hipLaunchKernelGGL(stream_write, dim3(32768), dim3(256), 0, 0, (float4*)bufs[i].ptr, 2 * half4, 1.0f);
                    else hipLaunchKernelGGL(stream_write2, dim3(32768), dim3(256), 0, 0, (float4*)bufs[i].ptr, (float4*)bufs[j].ptr, half4, 1.0f);
                };
                f();
                printf(" %5.2f", 2.0 * half4 * 16.0 / time_us(f, 3) / 1e6);
            }
            printf("\n"); fflush(stdout);
        }
    }
    // the remapped real kernels write the same bits as the real kernel
    if (!pmc) {
        double ref = 0.0;
        for (int k : {0, 2, 3, 12, 13, 14}) {
            vars[k].run((float*)bufs[0].ptr);
            CK(hipMemset(d_sum, 0, 8));
            hipLaunchKernelGGL(checksum_f4, dim3(2048), dim3(256), 0, 0, (const float4*)bufs[0].ptr, (size_t)NOUT * HW, d_sum);
            double s; CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
            if (k == 0) ref = s;
            printf("checksum %-7s %.17g %s\n", vars[k].name, s, s == ref ? "" : "(differs only by summation order if close)");
        }
        // correlation of every variant with the real kernel across the buffers
        const size_t n = bufs.size();
        for (size_t k = 1; k < vars.size(); ++k) {
            double mx = 0, my = 0; for (size_t b = 0; b < n; ++b) { mx += tab[b][0]; my += tab[b][k]; } mx /= n; my /= n;
            double sxy = 0, sxx = 0, syy = 0;
            for (size_t b = 0; b < n; ++b) { sxy += (tab[b][0] - mx) * (tab[b][k] - my); sxx += (tab[b][0] - mx) * (tab[b][0] - mx); syy += (tab[b][k] - my) * (tab[b][k] - my); }
            double lo = 1e9, hi = 0; for (size_t b = 0; b < n; ++b) { lo = std::min(lo, tab[b][k]); hi = std::max(hi, tab[b][k]); }
            printf("corr(k1, %-7s) = %+.3f   range %.2f .. %.2f us per frame\n", vars[k].name, (sxx > 0 && syy > 0) ? sxy / sqrt(sxx * syy) : 0.0, lo, hi);
        }
    }
    return 0;
}
