// VALU issue-cost micro-benchmark for gfx950 (not part of the product): cycles per wave-instruction of the
// instruction kinds the band kernel is made of, at 1/2/4/8 waves per SIMD.  Every test body is 64 independent
// instructions (8 accumulator sets) in a loop; cycles come from s_memtime (shader clock).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu.hip -o build_variants/valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

typedef float v2f __attribute__((ext_vector_type(2)));

enum { T_FMA, T_PKFMA, T_PKMUL, T_PKADD, T_MOV64, T_MOV32, T_DPPMOV, T_DPPFMAC, T_EXP, T_LOG, T_RCP, T_SQRT, T_MIN, T_MED3,
       T_CNDMASK, T_FLOOR, T_CVT, T_MUL, T_ADD, T_DSREAD128, T_DSREAD32, T_EXPFMA, T_PERMLANE_SWAP, T_COUNT };
static const char* names[T_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mov_b64", "v_mov_b32",
                                     "v_mov_b32_dpp wave_shr", "v_fmac_f32_dpp wave_shr", "v_exp_f32", "v_log_f32", "v_rcp_f32",
                                     "v_sqrt_f32", "v_min_f32", "v_med3_f32", "v_cndmask_b32", "v_floor_f32", "v_cvt_i32_f32",
                                     "v_mul_f32", "v_add_f32", "ds_read_b128", "ds_read_b32", "v_exp+3 v_fma (interleaved)",
                                     "v_permlane32_swap"};

template <int T>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    __shared__ float4 lds[256];
    lds[threadIdx.x & 255] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    float a[8], b[8];
    v2f p[8], q[8];
    float4 l4[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = 1.0f + 0.001f * (threadIdx.x + i);
        b[i] = 0.5f + 0.002f * i;
        p[i] = v2f{a[i], b[i]};
        q[i] = v2f{b[i], a[i]};
        l4[i] = make_float4(0, 0, 0, 0);
    }
    const float c0 = 0.999f, c1 = 1e-4f;
    const v2f pc = v2f{c0, c0};
    unsigned int addr = (threadIdx.x & 255) * 16;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(q[i]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define MOV64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(p[i]) : "v"(q[i]));
#define MOV32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define DPPMOV(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(b[i]));
#define DPPFMAC(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]), "v"(c1));
#define EXP(i) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define LOG(i) asm volatile("v_log_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c0));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define FLOOR(i) asm volatile("v_floor_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define CVT(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
#define DSR128(i) asm volatile("ds_read_b128 %0, %1" : "=v"(l4[i]) : "v"(addr));
#define DSR32(i) asm volatile("ds_read_b32 %0, %1" : "=v"(a[i]) : "v"(addr));
#define EXPFMA(i) asm volatile("v_exp_f32 %0, %1\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4" : "=v"(a[i]), "+v"(b[i]) : "v"(b[(i + 1) & 7] ), "v"(c0), "v"(c1));
#define PLSWAP(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            if constexpr (T == T_FMA) { REP8(FMA) }
            if constexpr (T == T_PKFMA) { REP8(PKFMA) }
            if constexpr (T == T_PKMUL) { REP8(PKMUL) }
            if constexpr (T == T_PKADD) { REP8(PKADD) }
            if constexpr (T == T_MOV64) { REP8(MOV64) }
            if constexpr (T == T_MOV32) { REP8(MOV32) }
            if constexpr (T == T_DPPMOV) { REP8(DPPMOV) }
            if constexpr (T == T_DPPFMAC) { REP8(DPPFMAC) }
            if constexpr (T == T_EXP) { REP8(EXP) }
            if constexpr (T == T_LOG) { REP8(LOG) }
            if constexpr (T == T_RCP) { REP8(RCP) }
            if constexpr (T == T_SQRT) { REP8(SQRT) }
            if constexpr (T == T_MIN) { REP8(MIN) }
            if constexpr (T == T_MED3) { REP8(MED3) }
            if constexpr (T == T_CNDMASK) { REP8(CNDMASK) }
            if constexpr (T == T_FLOOR) { REP8(FLOOR) }
            if constexpr (T == T_CVT) { REP8(CVT) }
            if constexpr (T == T_MUL) { REP8(MUL) }
            if constexpr (T == T_ADD) { REP8(ADD) }
            if constexpr (T == T_DSREAD128) { REP8(DSR128) asm volatile("s_waitcnt lgkmcnt(0)"); }
            if constexpr (T == T_DSREAD32) { REP8(DSR32) asm volatile("s_waitcnt lgkmcnt(0)"); }
            if constexpr (T == T_EXPFMA) { EXPFMA(0) EXPFMA(2) EXPFMA(4) EXPFMA(6) EXPFMA(1) EXPFMA(3) EXPFMA(5) EXPFMA(7) }
            if constexpr (T == T_PERMLANE_SWAP) { REP8(PLSWAP) }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + b[i] + p[i].x + p[i].y + q[i].x + l4[i].x + l4[i].w;
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int T>
static void run(float* out, long long* d_cyc, int cus) {
    const int iters = 4000;
    printf("%-28s", names[T]);
    for (int wps : {2, 4, 8}) {                        // waves per SIMD: 512-thread blocks = 2 waves on each of the 4 SIMDs
        const int blocks = cus * (wps / 2);
        std::vector<long long> h((size_t)blocks * 8);
        k<T><<<blocks, 512>>>(out, d_cyc, 10);
        hipDeviceSynchronize();
        k<T><<<blocks, 512>>>(out, d_cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= (double)h.size();
        const double ninstr = (double)iters * (T == T_EXPFMA ? 8 : 64);
        printf("  %d waves/SIMD: %6.2f cyc", wps, avg / ninstr / wps);
    }
    printf("\n");
}

template <int T>
static void run_all(float* out, long long* d_cyc, int cus) {
    run<T>(out, d_cyc, cus);
    if constexpr (T + 1 < T_COUNT) run_all<T + 1>(out, d_cyc, cus);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs; SIMD cycles per wave64 instruction = wave elapsed cycles (s_memtime) / instructions / co-resident waves; 512-thread blocks, all resident; EXPFMA row is per group of {1 exp + 3 fma}\n", prop.name, cus);
    float* out;
    long long* d_cyc;
    hipMalloc(&out, 1024);
    hipMalloc(&d_cyc, sizeof(long long) * cus * 4 * 8 * 8);
    run_all<0>(out, d_cyc, cus);
    return 0;
}
