// The small end of the pyramid in ONE launch.  Included by fvvdp_hip.hip after band_kernel.hpp and aux_kernels.hpp.
#pragma once
// ------------------------------------------------------------------------------------------------------------
// band_tail_kernel: the pyramid levels whose frames are small (<= TAIL_MAX_PX pixels), the pooled-sum finalisation of ALL
// bands and -- when the caller asks for it and this launch completes the clip -- the band / channel / frame pooling with the
// JOD regression (do_pooling_and_jods, fvvdp.py:337-357).  One workgroup of TAIL_WPT waves owns ONE frame and walks its
// levels in order: level i+1 is written by this workgroup and read back by it after a workgroup barrier (same CU), the
// frames run side by side on different CUs.  The per-item code is band_kernel's (band_item), so the numbers differ from the
// per-level launches only through the grouping of the partial sums (different chunk heights).
//
// STATUS: an experiment that did NOT pay (VERDICT r2 item 7 asked for it; profiles/r03_tail_launch.md has the numbers), kept
// behind FVVDP_BAND_TAIL=1 and covered by tests.  The idea was to replace latency-sized launches (4K x60: levels 3-6 42 + 16 +
// 7 + 6 us, finalize 4 us, pooling 10 us) by one.  What the measurements say:
//   * levels 3-6 in the tail, 16 waves per frame: 195 us per launch against 88 us for the four per-level launches.  A frame's
//     levels 3-6 are ~120 us of VALU work for ONE CU, and 60 frames occupy 60 of the 256 CUs; the per-level launches spread
//     every level over the whole chip.
//   * levels 5-6 only (120x68, 60x34) + finalize + pooling: 57 us with 16 waves (16 chunks per level: the 8 halo rows of a chunk
//     dominate), 49 us with 4 waves (one per SIMD: no other wave hides the load latency of the dependent row chain), against
//     26 us + the 10 us pooling launch.  End to end 4.63-4.68 ms per 4K x60 pair against 4.59-4.62 ms.
// The small levels are not launch-bound enough: a level is a chain of dependent row steps, and only the per-level launches
// run all frames' chains of ONE level on all CUs at once.
//
// Pooling: every workgroup publishes its frame's Q, then takes a ticket (device-scope atomic after a device-scope release
// fence); the workgroup that draws the last ticket acquires and pools all frames.  No workgroup waits for another one.
// ------------------------------------------------------------------------------------------------------------
#define TAIL_MAX_LEVELS 6
#define TAIL_MAX_PX 160000
#ifndef TAIL_WPT
#define TAIL_WPT 4                      // waves per workgroup: one per SIMD.  The work of a frame is VALU-bound on ITS CU, so more
                                        // waves only add halo rows (16 waves, 16 chunks per level: 57 us per 4K x60 launch)
#endif

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
