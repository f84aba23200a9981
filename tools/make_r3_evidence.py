#!/usr/bin/env python3
"""Assemble the round-3 experiment logs (gpurun_out/s*/, written by tools/r3_session*.sh on the GPU box) into tracked files
under profiles/: K1 placement modes, the tail-launch experiment, the foveated-kernel variants, the fusion front end."""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def rd(rel, drop=("amdgpu.ids", "fvvdp: level 0 allocation")):
    p = os.path.join(G, rel)
    if not os.path.exists(p):
        return "(not collected: %s)" % rel
    return "\n".join(l for l in open(p).read().rstrip().split("\n") if not any(d in l for d in drop))


def w(name, txt):
    open(os.path.join(P, name), "w").write(txt)
    print("wrote profiles/%s (%d bytes)" % (name, len(txt)))


w("r03_k1_placement.md", f"""# Round 3 -- the two modes of the temporal kernel (K1) and what controls them

K1 = `temporal_vec_kernel<8,4,0>` on the 4K x60 uint8 pair of bench.py (182.5 MB per frame: 6 B read + 16 B written per pixel).
VERDICT r2 weak 6 / item 4: the same kernel runs at 31-32 us per frame (5.8 TB/s) or 36-38 us (5.0 TB/s) "depending on where the
8 GB scratch lands physically".  What the experiments of this round add (all: HIP events inside the library, us per frame):

1. **The allocation API does not matter** (`tools/experiments/gpu_k1_placement.py`; `FVVDP_ALLOC=vmm` takes `hipMemCreate` / `hipMemMap` of the
   recommended granularity, `FVVDP_VMM_ALIGN_MB=1024` reserves the range 1 GiB-aligned):

```
{rd('s2/k1_placement.txt')}
```

2. **The virtual address does not matter.**  `tools/experiments/gpu_k1_offset_sweep.py`, first version: level 0 starts `offset` KB into an
   allocation that grows by the same amount -- the allocator grows it downwards, so most rows have level 0 at the SAME virtual
   address (0x...95000000) with a NEW physical allocation each time, and the time still varies between 32.9 and 38.2:

```
{rd('s4/k1_offsets.txt')}
```

3. **An offset inside ONE allocation does not matter** (second version: 1.1 GB of slack allocated once per context,
   `FVVDP_L0_OFFSET_KB` read at every call; K1/K2b; 4 KB steps would rotate any 8-way page interleave, 1-7 KB are misaligned
   to the 4 KB runs a wave writes):

```
{rd('s7/k1_fine.txt')}
```

   (`s5`, offsets 4 KB ... 1 GB on another box: 37.4-38.0 for every offset of every allocation.)

4. **It is a property of the allocation, and the pyramid kernel moves the other way** (`tools/experiments/gpu_k1_probe_corr.py`: 14
   re-creations of the scratch in one process; a cheap probe -- the same kernel fed from one dummy frame -- does NOT predict the
   mode of the real clip, i.e. the relation to the SOURCE buffers is part of it):

```
{rd('s6/k1_probe.txt')}
```

   K1 fast (31) goes with K2b 35.6-38.5, K1 slow (36) with K2b 34.3: the sum is 67.3 vs 70.8 us per frame.

5. **Consecutive allocations alternate strictly** (`tools/experiments/gpu_k1_clocks.py`: re-creation in a loop, 1.2 s of calls each; the clocks
   and the power the driver reports do not differ between the modes -- fclk 1250, mclk 2000 throughout; the sclk reading is the idle
   value between calls):

```
{rd('s8/k1_clocks.txt')}
```

6. **A short probe ranks the buffers the wrong way round** (first attempt at a fix: time 12 frames into the current buffer and
   into two fresh ones inside the second call, keep the fastest.  `tools/experiments/gpu_k1_probe_check.py`: the buffer that runs the full
   clip in 32.4 us per frame measures 38.5 in the probe, the 36.5 one measures 36):

```
{rd('s9/probe_on.txt')}
```

7. **Second attempt, kept as an opt-in (`FVVDP_PLACEMENT_PROBE=1`): online selection with the real calls as the measurement** (`temporal_channels_core` in `fvvdp_hip.hip`): the
   2nd full-batch call of a context is timed on the buffer it has, the 3rd runs -- and is timed -- on a freshly allocated one, the
   4th keeps the faster and frees the other; one more candidate if the two were within 5 %.  No extra GPU work, results
   independent of the buffer (asserted by the tool).  Per re-created context, K1/K2b call by call, selection on:

```
{rd('s10/probe_on.txt')}
```

   and off:

```
{rd('s10/probe_off.txt')}
```

   bench.py (2 warm-up + 10 timed steps, so the selection runs inside the timed region), separate processes, alternating:

```
{rd('s10/bench_ab.txt')}
```

   Final states with the selection: 34.6, 31.6, 31.7, 31.1, 31.1, 34.6, 34.1, 34.7 (mean 32.9); without: 36.6, 34.6, 31.8, 31.2,
   31.2, 36.4, 36.7, 34.4 (mean 34.1).  There are three modes (31.5 / 34.5 / 36.5), a candidate is as likely to be worse as
   better, the calls that run on a fresh buffer pay its first touch (39-43 us per frame), and the context holds level 0 twice
   for two calls.  A 3.5 % gain on K1 in the mean for that price: off by default.

8. **It is the physical FRAGMENTATION of the level-0 buffer, and the two kernels want opposite things** (later sessions).
   `FVVDP_ALLOC_FLAGS=contiguous` (`hipExtMallocWithFlags(hipDeviceMallocContiguous)`) puts K1 in its slow mode every time, the other
   flags do not decide it:

```
{rd('s16/alloc_flags.txt')}
```

   Building the buffer from physical chunks with the virtual-memory API (`FVVDP_ALLOC=vmm FVVDP_VMM_CHUNK_MB=c FVVDP_VMM_INTERLEAVE=k`:
   chunks created in order, mapped k-way interleaved): 2 MB chunks put K1 in its FAST mode almost every time -- and K2b, which
   reads the same buffer, at 38 us.  The sum K1 + K2b is 70-71 us per frame at both ends; 67 us (K1 fast AND K2b at 34.5) shows up
   in about one context in five with large chunks, not reproducibly:

```
{rd('s17/vmm_layout.txt')}
```

   A padded frame-slot stride (`tools/experiments/patches/level0_slot_stride_and_vmm_layout.patch`: every kernel that touches level 0
   takes a slot stride; 135 GPU tests green with a 68 KB pad) on contiguous memory -- is the slow mode a resonance of the
   132,710,400-byte frame stride?  No:

```
{rd('s18/slot_pad.txt')}
```

   Is it the page-table fragment size of the mapping?  Level 0 mapped twice with the virtual-memory API -- the readers use one
   mapping, K1 stores through a second mapping of the SAME physical allocation shifted by 2 ... 512 MB
   (`tools/experiments/patches/level0_write_alias.patch`): no change, K1 stays in the slow mode of a single physical allocation
   (`hipMemAddressReserve` does not honour a 1 GiB alignment request here, so the fragment sizes could not be forced either way):

```
{rd('s19/alias.txt', drop=("amdgpu.ids", "fvvdp: level 0 at"))}
```

   The order in which the XCDs walk the frame?  K1's workgroups take consecutive 1 KB pixel runs, so the eight XCDs (workgroup b runs
   on XCD b % 8) write interleaved runs; with each XCD on a contiguous eighth of the frame instead (`-DK1_XCD_ORDER=1`, the work order
   of the pyramid kernels; experiment at the end of the round, not kept): 37.8-38.8 us against 36.5-37.8 us on the same (slow-mode) box.

What it is physically is not known to me: not address translation (r2: UTCL1 / UTCL2 counters equal in both modes), not the
clocks, not the alignment of the buffer, not an XCD <-> page interleave (a 4 KB offset changes nothing).  The per-channel HBM
counters that would show a channel imbalance are not exposed by rocprofv3 on this stack (TCC_EA0_* are aggregates).
""")

w("r03_tail_launch.md", f"""# Round 3 -- one launch for the small pyramid levels + finalisation + pooling: measured, not kept as default

VERDICT r2 item 7 asked for levels >= 3, `finalize_kernel` and `pool_jod_kernel` in one launch.  `band_tail_kernel`
(`csrc/tail_kernel.hpp`; one workgroup per frame walks its levels in order, the last workgroup to finish pools) exists, is
tested (`tests/test_gpu_fused.py`) and stays behind `FVVDP_BAND_TAIL=1`; `fvvdp_bands_forward_pool` is the entry point.
bench.py (4K x60, 10 steps), `levels` = HIP-event us per frame of levels 0..6 (a merged level reports at its first level):

levels 3-6 + finalize + pooling in the tail, 16 waves per workgroup (tail=1) against the per-level launches (tail=0):

```
{rd('s3/tail_ab.txt')}
```

levels 5-6 + finalize + pooling only, 16 waves (tail=d) against none:

```
{rd('s4/tail_ab.txt')}
```

the same with 4 waves per workgroup (one per SIMD; wpt4) and 16 (wpt16):

```
{rd('s5/tail_ab.txt')}
```

Still images (tail default-on at that point vs FVVDP_BAND_TAIL=0):

```
{rd('s5/image_probe.txt')}
```

Why: a frame's levels 3-6 are ~120 us of VALU work for ONE CU (16 waves share its 4 SIMDs), and 60 frames occupy 60 of 256 CUs;
the per-level launches spread every level of every frame over the whole chip (88 us for the four).  For the two tiny levels the
chain of dependent row steps of one frame (17 + 13 steps, nothing to hide the load latency with one wave per SIMD) is longer
than two launch latencies.

A last variant at the end of the round: an AUTOMATIC policy -- the tail takes only the levels from which on the pyramid holds at most
`FVVDP_TAIL_PX` pixels per frame (+ finalize + pooling); `tools/experiments/r3_session30.sh`, `tools/gpu_small_latency.py`, bench.py, 1080p x60:

```
{rd('s30/tail_auto.txt')}
```

A 512x512 image: 10 launches -> 4, GPU-side 96.6 -> 81-83 us per call, but `predict()` as the caller sees it (with its host
synchronisation) 129 -> 128-155 us: the call is bound by the ~100 us of host code between two calls (rocprofv3 time line: kernels of
4-6 us with 4-6 us gaps, a 30 us tail launch, then > 100 us without any kernel).  The 4K x60 pair and 1080p x60: within noise
(4.29-4.36 vs 4.35-4.51 / 4.38 / 4.28-4.36 ms; 1.38 vs 1.36-1.39 ms).  Not made the default; the opt-in stays as it was.
""")

w("r03_fov_variants.md", f"""# Round 3 -- foveated kernel (BASELINE configs[3]): what bounds it, and the variants that were measured

**Superseded reading.**  The "Reading" paragraph at the end was written from the SQ counters and concluded "VALU-bound".  Ablation builds
made later in the round (`profiles/r03_pyramid_bounds.md`) showed that neither the instruction count nor the byte count bound this
kernel: every step waited for HBM behind the in-order load queue (map records requested after the row prefetch).  With that fixed the
kernel runs 40.5-41.3 us at level 0 and 54.4 us over all levels.  The measurements below (all made BEFORE the fix) stand as measurements,
their flatness is explained by that stall; the counter-based conclusion does not stand.

`tools/gpu_config4.py` (4K x120, standard_hdr_pq, moving gaze), kernel us per frame = [K1, level 0, level 1, ...]:
default = LUT slice padded to 36-entry rows (bank-conflict-free, `FOV_ROW` in band_kernel.hpp); nopad = the round-2 layout;
pad33 = 33-entry rows; wpb6 / wpb12 = 6 / 12 waves per workgroup sharing the slice; phase4 = all four pixels of a step batched.

```
{rd('s2/fov_ab.txt')}
```

SQ / TCC counters of the round-2 layout (`tools/pmc_sq_summary.py`; levels 0 and 1 have the same grid and are averaged in the
first block: 6 dispatches = 3 calls x 2 levels):

{rd('s1/pmc_sq_fov_bandonly.md')}

With the padded slice (LDS conflict cycles / LDS active cycles 0.148 -> 0.065 at levels 0+1, 0.425 -> 0.146 at level 2; the time
does not move: LDS is 1.3 % of the waves' time):

{rd('s2/pmc_fov_padded.md')}

Two more variants, measured after the first version of this file:

```
{rd('s14/fov_lane_eff.txt')}
```

`strip54`: every strip advances by 54 instead of 60 columns (timing only: neighbouring strips overlap, the sums are wrong on
purpose) -- 12.5 % more waves of unchanged work, the lane efficiency a two-level strip would have.  Levels >= 1 get 10-12 % slower
(10.1 -> 11.3 us); level 0 does not move, but its wave count changes from 5.0 to 5.6 "rounds" of the chip, so that row is not
conclusive on its own.  `norhomap` (`FVVDP_FOV_NO_RHOMAP=1`: the rho-axis position recomputed per pixel and frame, no map read):
56.8 us at level 0 -- the bytes saved do not pay for the instructions.

The rho map at 4 instead of 8 bytes per pixel (interval + fraction in one float, unpacked with floor / sub / mul: +3 VALU
instructions per pixel, -41 MB per frame at levels 0+1; `tools/experiments/patches/fov_compact_rho_map.patch`, all 157 GPU tests
green with it), three runs:

```
{rd('s15/fov_probe.txt')}
```

45.1-45.9 us at level 0 against 44.3-44.7, 10.5 against 10.1-10.3 at level 1: 14 % fewer bytes at level 0 buy nothing, 3 % more
instructions cost 1.5 % -- the kernel is bound by its instruction stream, not by the bytes.  Not kept.

Reading: `valu` 0.318 of a wave's resident time x 3 waves per SIMD = 0.95 -> the VALU pipe is busy 95 % of the time (the plain
two-level kernel: 0.357 x 3 = 1.07, saturated, `profiles/r03_final_kernel_trace.md`); at the same time the two levels move
254 MB per frame (level 0: 132.7 MB + 66 MB of rho map + 33 MB written, plus halo) in ~55 us = 4.6 TB/s, level 0 alone 240 MB in
45 us = 5.3 TB/s -- the memory system's ceiling for a read/write mix (5.0-5.6 TB/s, `tools/microbench/mix.hip`).  Both limits are close, but the
experiments above decide it: fewer bytes change nothing, more instructions cost time.  Removing instructions is bounded by the 13
transcendentals and the trilinear blend per pixel.  A two-level variant (VERDICT r2 item 2a) multiplies the VALU work by
60/54 (strip halo) on a kernel that is already VALU-bound and saves 17 % of the bytes: 58 us against 55 by the same arithmetic
that predicted the plain kernel's gain.  Target (<= 50 us for all levels) not met: 59.5 us.
""")

w("r03_fusion_front_end.md", f"""# Round 3 -- K1 -> K2 fusion: the front end of the one variant that had not been costed on silicon

VERDICT r2 item 9: (strip, chunk)-persistent single-wave workgroups that walk the frames of their tile in order and re-compute the
temporal channels from the uint8 frames through L2 / Infinity Cache, writing no level 0.  `tools/microbench/fuse_front.hip` runs
ONLY that front end (unpack, LDS table, RGB -> Y, 8-tap FIR of both streams; checksum instead of a pyramid): 4 pixels per
lane, `rows` owned rows + `halo` rows per tile and frame, every tile walks all 60 frames.  If this alone costs more than K1 (35 us
per frame) plus the hand-off it removes (level 0 written + read: 265 MB = ~50 us at the mix ceiling), the fused kernel -- this
front end in front of K2b's VALU-bound 34 us in the SAME wave -- cannot win.

```
{rd('s8/fuse_front.txt')}
```

rocprofv3 --kernel-trace --stats of the first line:

{rd('s8/fuse_front_trace.md')}

The front end alone takes 260-740 us per frame.  Two things bind: tile-persistent workgroups are few (510 waves at 64-row
tiles; 3000+ waves only at <= 10-row tiles, where the 14 halo rows of the two-level pyramid make the tile 2.4x its owned rows),
and the 8-frame window costs 48 dword loads + 48 LDS look-ups + 32 multiply-adds per 4 pixels and frame in place of K1's 6 + 6 + 32
(the ring in registers converts every source frame once).  Not pursued further; the hand-off stays.
""")

w("r03_yuv_ingest.md", f"""# Round 3 -- what bounds the planar-YUV ingest kernel (VERDICT r2 item 3: 10-bit 4:2:0 at 60 fps <= 50 us per frame)

`temporal_yuv_vec_kernel<16, uint16, 4:2:0, sRGB>` on a 3840x2160x60 pair at 60 fps (16-tap window), `tools/gpu_yuv.py`.

Round-3 steps, us per 4K frame (rocprof steady median / 60): 65.2 (round 3 start, no spills, 2 waves per SIMD) -> **63.0** with the
window kept in place (frame v in slot v mod 16, one straight-line FIR per slot behind a wave-uniform switch; before, the 16-slot
window was shifted by one slot per frame = 120 `v_mov_b32` of 860 VALU instructions per lane and frame).  The 8-slot kernels (30 fps)
do not change (46 us 4:2:0 8 bit, 37 us 4:4:4 8 bit).

```
{rd('s20/yuv_probe.txt')}
```

{rd('s20/kernel_trace_yuv.md')}

SQ counters of the same call (`tools/experiments/r3_session21.sh`: five `--pmc` passes, `tools/pmc_sq_summary.py`):

{rd('s21/pmc_sq_2160_3840_60_10_420_60.md')}

{rd('s21/pmc_sq_2160_3840_60_8_420.md')}

Reading.  16-slot kernel: a wave issues 47.1 k VALU instructions over its 75 frame steps = **628 per step of 4 pixel pairs** (128 of them
the FIR: 16 taps x 4 pixels x {{sustained, transient}}; ~440 the conversion: 32 integer -> float, the 4:2:0 bilinear blend, the
colour matrix, 24 `v_log_f32` + 24 `v_exp_f32` of the sRGB curve, the luminance sum with the reference's roundings).  With 2 waves per SIMD
(the 128-register window leaves no room for a third) the VALU is busy 0.44 x 2 = **0.88** of the time, 0.20 of a wave's time is
`s_waitcnt` (the tap loads at the head of every FIR variant, the LDS transpose), the rest is the other wave's VALU.  The kernel
moves 5.7 + 8.0 GB in 3.8 ms = 3.6 TB/s: not the memory system.  50 us would need ~490 VALU instructions per step at the same
occupancy; the arithmetic of the path (3 transcendental pairs per pixel and stream, 256 multiply-adds of FIR per 4 pixels) does not
go below ~600 without changing the reference's roundings.  Target not met; the kernel is VALU-bound.

Cross-check by ablation (`-DYUV_ABLATE_MEM`: every wave converts the same 64 pixel quads of source frame 0 and stores nothing;
`tools/experiments/r3_session29.sh`, us per frame in the last column; JOD 10 = test and reference read the same data):

```
{rd('s29/yuv_ab.txt')}
```

The arithmetic alone takes 55 us (16 slots, 10 bit 4:2:0), 39.5 us (8 slots, 4:2:0) and 32 us (8 slots, 4:4:4) of the 63.6 / 46.3 / 37.5 us
of the full kernels: arithmetic-bound with ~15 % lost to overlap at 2-3 waves per SIMD.  (Also tried: the ten source loads of a frame
pair in a fixed order, which turns the `s_waitcnt vmcnt(0)` behind the four stores at the top of every step into counted waits:
63.0-63.2 against 62.8-63.6 us, not kept.)
""")

w("r03_pyramid_bounds.md", f"""# Round 3 -- what bounds the pyramid kernels: ablation builds, not counters

`tools/experiments/r3_session27.sh`; builds by `tools/build_variant.sh <name> "<flags>"`; us per 4K frame (HIP events inside the
library, median over the calls of one process; three processes per build, alternating).  The SQ counters of
`profiles/r03_final_kernel_trace.md` (a wave of `band2_kernel<4>` issues VALU work 0.357 of its resident time, x 3 waves per SIMD
= 1.07) were read as "VALU-bound" earlier in this round.  The ablations say the kernel is **co-bound**:

## Two-level kernel (`band2_kernel<4>`, levels 0+1 of the bench pair)

```
{rd('s27/band2_bounds.txt')}
```

* `default` 35.1-35.6 us (this box; 32.9-34.4 on others).
* `b2abl` (`-DBAND2_ABLATE=1`: no per-pixel tail -- no contrast, CSF, masking, pooling; loads, both reduce / expand filters and the
  level-C store stay): 29.2-31.6 us.  That is 153.9 MB per frame (PMC) in 30 us = 5.1 TB/s, the memory system's ceiling for this
  mix (`tools/microbench/mix.hip`: 5.0-5.6 TB/s) -- the **memory floor**.  4 waves per SIMD instead of 3 (`b2abl4`, 33 % more bytes
  in flight) do not lower it: 30.0-31.8 us.  More prefetch depth would not either.
* `b2nomem` (`-DBAND2_ABLATE_MEM`: ALL arithmetic, but every wave re-reads the same 8 L2-resident rows and stores nothing):
  30.5-30.9 us -- the **arithmetic floor**.
* Both floors are 30 us; together the kernel takes 33-35.6 us: 10-18 % of imperfect overlap between three waves per SIMD that each
  alternate between a row wait and ~1600 cycles of arithmetic.  The registers (163 of 168) leave no room for a fourth wave or a
  second step of prefetch.  Removing arithmetic alone cannot win more than the overlap loss: removing the whole tail (more than half
  of the instructions) gains 14 %.

## One-level kernels (`band_kernel<4,false,0|1>`)

```
{rd('s27/band1_bounds.txt')}
```

```
{rd('s27/fov_bounds.txt')}
```

* Non-foveated level 0, one level per launch: 32.3-36.8 us with the tail, 30.2-35.1 us without: the data flow (166 MB per frame at a
  4:1 read:write mix) is the bound, the arithmetic hides behind it.
* Foveated level 0 (second number of the `kernel us/frame` lists; first = K1): 40.5-41.8 us with the tail, 30.6-34.8 us without,
  39.7-41.8 us without the eccentricity arithmetic (`noecc2`: 183 of 1928 VALU instructions of the kernel, all 32 `v_sqrt_f32`): the
  instruction COUNT is not the bound here either.  What was: every step waited for HBM.  The rho-map records of a step were
  requested after the two prefetched rows of the next step, and loads return in order (`s_waitcnt vmcnt`), so the wait for the map was
  a wait for rows requested ~1000 cycles earlier.  Requesting the map first (and walking the frames of a tile back to back so that its
  map slice stays in the XCD's L2: HBM reads of the level-0 launch 3829 -> 2683 MB per 18 frames, PMC) brought level 0 from 44.7 to
  40.6 us and the whole foveated pass from 59.3 to 54.4 us per frame:

```
{rd('s28/ab.txt')}
```
(`tools/experiments/r3_session28.sh`: `prev` = the band kernels of commit 005ba66 -- shifted row window, map loads behind the row prefetch,
strip-fastest work order; `nofirst` = this build with `-DFOV_FRAME_FASTEST=0`; `default` = this build.  Per line: wall time of the
4K x120 foveated call, then HIP-event medians [K1, level 0, level 1, ...].  The row-window ring alone gave 2 %, the load order 5-6 %,
the work order nothing measurable in time -- it removes 30 % of the kernel's HBM reads.)

Where the remaining 5-6 us above the data-flow floor go (`-DFOV_ABLATE_LDS`: the four LUT-cell reads per pixel replaced by constants;
`abl1` = no per-pixel tail; same box, alternating):

```
{rd('s31/fov_nolds.txt')}
```

The LDS reads are worth ~1 us; the rest is the same imperfect overlap of arithmetic and memory as in the two-level kernel.

The arithmetic floor of the one-level kernels (`-DBAND_ABLATE_MEM`: all arithmetic, rows re-read from 8 L2-resident rows, nothing stored;
`b1nomem`), next to their data-flow floor (`abl1`), same box, alternating; foveated (second number = level 0), then plain with
`FVVDP_BAND_FUSE=0` (first number = level 0):

```
{rd('s33/fov_floors.txt')}
```

```
{rd('s33/one_level_floors.txt')}
```

Foveated level 0: arithmetic alone 33.0-33.2 us, data flow alone 31.4-35.2 us, together 40.4-41.3 us -- co-bound like the two-level
kernel, with a larger overlap loss (more wait points per step: map, two LDS phases).  A two-level foveated variant would have an
arithmetic floor of (33 + 8) x 60/54 = 45.6 us for levels 0+1 against 50.1 us for the two launches today: no room for the 8-10 us the
target needs.  Plain one-level level 0: arithmetic 22.5 us under a data flow of 30 us (32.2 together): memory-bound, which is why
the two-level kernel pays there.

```
{rd('s26/fov.txt')}
```
{rd('s26/pmc.txt')}
""")
