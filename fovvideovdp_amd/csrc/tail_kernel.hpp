// The small end of the pyramid in ONE launch.  Included by fvvdp_hip.hip after band_kernel.hpp and aux_kernels.hpp.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// band_tail_kernel: pyramid levels whose frames are small (<= TAIL_MAX_PX pixels: at 4K levels 3..6), the pooled-sum
// finalisation of ALL bands and -- when the caller asks for it and this launch completes the clip -- the band / channel /
// frame pooling with the JOD regression (do_pooling_and_jods, fvvdp.py:337-357).
//
// Why: each of these was a launch of its own (4K x60: levels 3-6 42 + 16 + 7 + 6 us, finalize 4 us, pooling 10 us, and a
// few microseconds of gap between dependent launches): latency-sized kernels that cannot fill the chip.  Here one workgroup
// of WPT waves owns ONE frame and walks its levels in order -- level i+1 is written by this workgroup and read back by it
// after a workgroup barrier (same CU: the data goes through its own L1 / the XCD's L2), the frames run side by side on
// different CUs.  The per-item code is band_kernel's (band_item), so the numbers differ from the per-level launches only
// through the grouping of the partial sums (different chunk heights).
//
// Pooling: every workgroup publishes its frame's Q, then takes a ticket (device-scope atomic after a device-scope release
// fence); the workgroup that draws the last ticket acquires and pools all frames.  No workgroup waits for another one.
// ------------------------------------------------------------------------------------------------------------
#define TAIL_MAX_LEVELS 6
#define TAIL_MAX_PX 160000
#define TAIL_WPT 16                     // waves per workgroup (band_item<4, false, 0>: 124 VGPRs -> 4 waves per SIMD)

struct TailArgs {
    BandArgs band[TAIL_MAX_LEVELS];
    int n_levels;
    FinalizeArgs fin;
    int do_pool;
    PoolArgs pool;
    unsigned int* ticket;               // zero between launches (the pooling workgroup resets it)
};

static_assert(sizeof(TailArgs) <= 4096, "kernel arguments are limited to 4 KB");

template <int P>
__global__ __launch_bounds__(64 * TAIL_WPT) void band_tail_kernel(const TailArgs t_byval) {
    // the argument block is indexed with a run-time level number: read it from the kernel-argument segment (scalar loads)
    // instead of a private copy
    const TailArgs& t = *(const TailArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)t_byval;
    __shared__ float4 s_csf[FVVDP_LUT_N];
    __shared__ float2 s_ax[1];
    __shared__ double s_part[256];
    __shared__ unsigned int s_ticket;
    const int frame = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    for (int li = 0; li < t.n_levels; ++li) {
        const BandArgs& a = t.band[li];
        __syncthreads();                // the previous level of this frame is complete (and s_csf is free again)
        band_load_tables<0>(a, s_csf, s_ax, (int)threadIdx.x, 64 * TAIL_WPT);
        __syncthreads();
        const int items = a.n_strips * a.n_chunks;
        for (int it = wave; it < items; it += TAIL_WPT)
            band_item<P, false, 0>(a, it % a.n_strips, it / a.n_strips, frame, lane, s_csf, s_ax);
    }
    __syncthreads();                    // every partial sum of this frame is written
    for (int pr = wave; pr < t.fin.n_bands * 2; pr += TAIL_WPT) finalize_one(t.fin, pr >> 1, pr & 1, frame, lane);
    if (!t.do_pool) return;
    __threadfence();                    // this frame's Q before the ticket (device scope)
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(t.ticket, 1u);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    __threadfence();                    // acquire: the other workgroups' Q
    if (threadIdx.x == 0) *t.ticket = 0u;
    pool_jod_body(t.pool, s_part, (int)threadIdx.x);
}
