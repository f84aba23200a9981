#!/usr/bin/env python3
"""Round 4: turn the logs of the GPU sessions (gpurun_out/r4s*/, written by tools/experiments/r4_session*.sh; one gpurun call =
one MI355X box) into the tracked evidence files under profiles/.  Every number in those files is copied from a log."""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def lines(sess, pat):
    rows = []
    for f in sorted(glob.glob(os.path.join(G, sess, pat))):
        try:
            d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        except Exception:
            rows.append((os.path.basename(f), None))
            continue
        rows.append((os.path.basename(f), d))
    return rows


def table(rows, extra=()):
    out = ["| run | ms per pair (median step) | K1 us/frame | levels 0+1 us/frame | all levels + finalize us/frame | JOD |" + "".join(" %s |" % e[0] for e in extra),
           "|---|---|---|---|---|---|" + "---|" * len(extra)]
    for name, d in rows:
        if d is None:
            out.append("| %s | failed | | | | |" % name)
            continue
        g = d.get("graded_pass", {})
        out.append("| %s | %.3f | %.1f | %.2f | %.2f | %s |" % (name.replace(".json", ""), d["ms_per_pair"], g.get("temporal_us_per_frame_median", 0),
                                                              g["levels_us_per_frame_median"][0], g.get("us_per_frame_all_levels", 0), d["jod"][0]) +
                   "".join(" %s |" % e[1](d) for e in extra))
    return "\n".join(out)


def rd(path):
    p = os.path.join(G, path)
    return open(p).read().rstrip() if os.path.exists(p) else "(not collected)"


def write(name, txt):
    with open(os.path.join(P, name), "w") as f:
        f.write(txt.rstrip() + "\n")
    print("wrote profiles/" + name)


# ---------------------------------------------------------------------------------------------------------------------------
write("r04_level0_chunks.md", """# Round 4: the level-0 scratch mapped from physical chunks (what the "placement modes" of rounds 2-3 were)

Rounds 2 and 3 found the temporal kernel K1 (unpack + display model + luminance + 8-tap FIR -> pyramid level 0; 6 B read, 16 B written
per pixel and frame) running in a fast (~31 us per 4K frame) or a slow (~37 us) mode that belonged to the physical allocation of the
level-0 scratch, and an online selection between two `hipMalloc` buffers that did not help (`profiles/r03_k1_placement.md`).  Round 4:

1. The selection was rebuilt on the SUM temporal + pyramid time with warm candidates (VERDICT r3 item 3) and measured on three boxes
   (sessions 1 and 3 below): on a box with free memory BOTH `hipMalloc` buffers are in the slow mode (K1 37.4-38.1 us in all 12
   processes of session 3, with or without the selection) -- there is nothing to select.
2. What decides the mode on most boxes is whether the buffer is ONE physically contiguous range.  `hipMalloc` of 8 GB on a fresh box returns one;
   the same virtual range mapped from physical chunks (`hipMemCreate` x n, `hipMemMap`) of 2 ... 128 MB puts K1 in the fast mode
   on those boxes every time (two of the ~14 boxes of the round stay slow in every allocation mode: last section), and the pyramid kernel that reads the buffer does not change (session 4; 512 MB chunks: half way).
3. The library now maps every pyramid level of >= 256 MB from 32 MB chunks (>= 16 MB from 4 MB chunks since the microbenchmark of the last section) (`vmm_alloc` in `csrc/fvvdp_hip.hip`;
   `FVVDP_ALLOC=malloc` goes back); the selection between two hipMalloc buffers and its C entry point were removed (a choice between the
   two KINDS came back at the end of the round: sessions 17-18 below).  Same-box A/B of the bench line
   (session 5): 4.57 -> 4.25 ms per 4K x 60 pair, bit-identical results (`test_chunk_mapped_scratch_gives_the_same_results_as_hipmalloc`).
   Creating the 4K x 60 context (about 330 chunks of 32 MB) and touching it for the first time: 20 ms (session 5, last lines).

## Session 3 (one box): the sum-based selection, `b_s1` = on, `b_s0` = off; `q8_*` = 8 pairs queued per step

`tools/experiments/r4_session3.sh`.  K1 / levels: HIP-event medians of the isolated kernels of the same process.

%s

The selection compared 75.5-76.3 (incumbent) with 74.4-75.0 us per frame (candidate) in every process: both buffers in the same mode.

## Session 4 (another box): the scratch through the virtual-memory API, chunk size; `default` = hipMalloc, `contig` = hipDeviceMallocContiguous

`tools/experiments/r4_session4.sh` (the build of that session still took `FVVDP_ALLOC=vmm FVVDP_VMM_CHUNK_MB=<n>` from the environment).

%s

## Session 5 (a third box): the shipped default (32 MB chunks) against `FVVDP_ALLOC=malloc`; 4K x 60, 8 queued pairs, 1080p x 60

`tools/experiments/r4_session5.sh`; the whole GPU test suite ran first in the same call (167 passed).

%s

## Session 9 (a fourth box): the 120-frame context of configs[3] (16 GB of level 0), `tools/gpu_config4.py`, kernel us per frame [K1, levels 0.., finalize]

```
%s
```

## Later sessions (15, 16 and three more boxes): not every box has the fast mode

`tools/experiments/r4_session15.sh` / `r4_session16.sh`; `stride*` = the chunks mapped in a permuted order (virtual slot i <- physical chunk
i * stride mod n), `linear*` = creation order, the digit = chunk size in MB, `src_expandable` = the SOURCE arrays in chunk-mapped memory as
well (`PYTORCH_HIP_ALLOC_CONF=expandable_segments:True`), `l0_malloc` = `FVVDP_ALLOC=malloc`.  ms per pair, K1 / levels 0+1 / all levels us per frame.

Session 15 -- a box on which K1 takes 37.4-38.5 us per frame in EVERY allocation mode, although its hipMalloc memory is of the
fast-write kind (microbenchmark below the table: 7.09 TB/s write-only on hipMalloc).  Neither the mapping order nor 2 MB chunks change it:

```
%s
%s
```

Session 16 (last call) and three further boxes:

```
%s
%s
```

Over the boxes of the round (shipped default, 32 MB chunks): K1 30.6, 31.6, 32.0-32.9, 32.8, 33.3, 33.3, 33.8, 37.1, 37.5 us per frame;
with hipMalloc: 37.0-38.1 on nine boxes, 35.1-36.0, 32.9, 32.0 on three.  The chunk-mapped scratch is faster on most boxes, equal on some,
4 %% slower on one, and there are boxes where nothing helps; what those boxes have in common is not known to me (same kernel, pyramid
kernels and streaming microbenchmarks at their usual rates).  The permuted mapping order was not kept.

## Sessions 17 and 18 (ten boxes): an online choice between the two kinds -- kept, on by default

Since neither kind is the fast one everywhere, a video context whose level 0 holds >= 1 GiB now compares them on its first eight
full-size calls (`selection_step` in `csrc/fvvdp_hip.hip`, `fvvdp_ctx_alloc_info`; two warm-ups, two timed calls on the buffer it
has, one untimed and two timed calls on a fresh buffer of the OTHER kind, the smaller of the two timings counts, the candidate has to
win by 3 %%) and keeps the faster; `FVVDP_PLACEMENT_PROBE=0` turns it off.  The timed quantity is the temporal kernel + the whole
pyramid pass of the batch (VERDICT r3 item 3).  `choice` = default, `vmm_fixed` / `malloc_fixed` = comparison off, `choice_from_malloc` =
starting from hipMalloc; ms per pair, K1 and levels 0+1 us per frame (HIP-event medians of the same process), then start -> kept
and the two timings in us per frame.  `tools/experiments/r4_session17.sh` / `r4_session18.sh`, one gpurun call per box.

First form (one timed call per kind, 1.5 %% margin; boxes 1-7).  It finds the faster kind where the kinds differ (box 4 from hipMalloc:
4.35 -> 4.04 ms; box 5 from chunks: 4.37-4.46 -> 4.17-4.26), is neutral where they do not (boxes 1, 3, 6), and mis-picked once
(box 4, first line: 71.27 vs 69.75 us measured, then 3 %% slower than `vmm_fixed`):

```
%s
```

Shipped form (two timed calls per kind, 3 %% margin; three more boxes): no mis-pick in 12 processes, gains of 2-4 %% where it switches.
A candidate allocated late in the life of the process is not always as fast as the same kind allocated first (box 9, `choice_from_malloc`
first run: 69.83 against 71.94 us is inside the margin, while `vmm_fixed` is 4 %% faster than `malloc_fixed` there) -- the choice removes
the bad cases, it does not always find the best one:

```
%s
```

## Session 45: eight allocations held at once in ONE process -- every allocation is its own draw

`tools/experiments/gpu_alloc_draws.py`: eight contexts (each with its own scratch) alive at the same time, the same 4K x 60 pair through
each, comparison off; temporal kernel / levels 0+1 us per frame (HIP events, median of 3 calls; two rounds over the contexts):

```
%s
```

A context keeps its speed from round to round, and the contexts of one process differ as much as boxes do: K1 31.3-38.3 us per frame
with either kind of allocation.  The pyramid kernel reading the same buffer tends the OTHER way (the context with the slowest K1, 38.2,
has the fastest levels 0+1, 28.4), so the sums spread less: 61.2-66.6 us per frame.  The kind of allocation shifts the odds; what decides
is where the particular buffer landed.  The context's online choice is therefore a best-of-two draw on the sum; a best-of-N over more
fresh buffers would gain about another 1-2 %% at 3 calls per candidate -- not built.

## Session 46: fewer resident waves do not cure the slow mode

The temporal kernel's occupancy capped through unused dynamic LDS (experiment build; 16 single-wave workgroups per CU = 4 per SIMD is the
shipped state), three contexts per process, K1 us per frame; boxes 1 and 2 had every context in the slow mode, box 3 in the fast one:

```
# box 1
%s
# box 2
%s
# box 3
%s
```

Slow mode: -1.5 %% at 8 per CU, -4 %% at 6 per CU; fast mode: +-0 down to 6 per CU, then steeply worse.  A cap at 8 per CU would be worth
0.3 us per frame at best; not built.

`tools/microbench/chunks.hip` on a box whose hipMalloc returns one contiguous range, TB/s -- it is the WRITES that such a range slows down
(write only 5.8-6.0 -> 6.9-7.1, copy 5.3-5.4 -> 6.0-6.1, reads unchanged); since then every level of >= 16 MB is chunk-mapped:

```
%s
```
""" % (table(lines("r4s3", "*.json")), table(lines("r4s4", "*.json")), table(lines("r4s5", "*.json")), rd("r4s9/fov_chunks.txt"),
       rd("r4s15/order.txt"), rd("r4s15/chunks_microbench.txt"), rd("r4s16/src.txt"), rd("r4s16/three_more_boxes.txt"),
       "\n".join("# box %d\n" % i + rd("r4s17/choice_box%d.txt" % i) for i in range(1, 8)),
       "\n".join("# box %d\n" % (7 + i) + rd("r4s18/choice2_box%d.txt" % i) for i in range(1, 4)),
       rd("r4s45/draws.txt"), rd("r4s46/pads_box1.txt"), rd("r4s46/pads_box2.txt"), rd("r4s46/pads_box3.txt"),
       rd("r4_chunks_microbench.txt")))

# ---------------------------------------------------------------------------------------------------------------------------
write("r04_stage_overlap.md", """# Round 4: temporal kernel of batch b+1 against the pyramid pass of batch b on two streams (VERDICT r3 item 1b) -- built, measured, off

Built: `fvvdp_bands_forward_at` (C ABI: the pass on level-0 slots [slot0, slot0 + n)), two banks of the level-0 scratch, two internal
non-blocking streams with events between them (`fvvdp.pipeline`, `_predict_on_device`), `fvvdp.predict_batch` (a queue of pairs whose
caller's stream joins once, so that pair p+1's temporal kernel is not held behind pair p's last pass).  Parity:
`tests/test_gpu_state.py::test_stage_overlap_equals_the_sequential_order` (bit-identical to the same batches run one after the other).

Result: **no gain in the product path** -- a single 4K x 60 pair in two halves 4.50-4.68 ms against 4.25-4.28 ms sequential; a queue of
8 pairs 4.30-4.34 against 4.12-4.38 ms per pair.  Both stages are bound by the same HBM (K1 moves 182.5 MB, the pass 167 MB per frame:
~350 MB in 67-70 us = 5.0-5.2 TB/s back to back, which IS what this memory system delivers for such mixes); side by side they take as
long as one after the other.  The sequential order stays the default (`pipeline = 0`).

## The isolated experiment of round 3, repeated with banks (`tools/experiments/gpu_coexec2.py`, one box)

```
%s
```

The "gain" of the two-stream rows is measured against the SUM of two separately synchronised timings, each of which carries its own
launch + synchronisation cost (~50 us); in the product path the sequential order has no such cost between the stages.  Launch order
matters: the pass first, the temporal kernel joining, is 3-6 %% better than the other way round (the pass keeps its three waves per
SIMD); the product path enforces that order with an event.

## Session 1 (one box): `b_p<k>_s<probe>_<rep>` = a single pair cut into k batches (p0 = sequential), natural launch order

%s

## Session 2 (another box): launch order, stream priorities, CU masks (`hipExtStreamCreateWithCUMask`: K1 on k of the 32 CUs of every XCD, the pass on the rest / on all)

`*_q4` = 4 pairs queued per step.  `k2first` = the pass is released first; `prio_k1` / `prio_k2` = that stream has the higher priority;
`cu<k>` = K1 on k CUs per XCD and the pass on the other 32-k; `cu<k>_all` = the pass on all CUs.

%s

K1 needs the whole chip to reach its bandwidth (on 16 of 32 CUs per XCD the pair takes 5.0 ms): a static partition loses more than
the overlap could win.

## Session 3 (a third box): queue of 8 pairs through `predict_batch` (the caller's stream joins once), `q8_p0` sequential, `q8_p2` / `q8_p3` two / three batches per pair with the pass released first, `q8_p2nat` natural order

%s
""" % (rd("r4_coexec2.txt"), table(lines("r4s1", "b_*.json")), table(lines("r4s2", "*.json")), table(lines("r4s3", "q8*.json"))))

# ---------------------------------------------------------------------------------------------------------------------------
write("r04_pyramid_kernel.md", """# Round 4: the two-level pyramid kernel (levels 0+1 of the bench pair) -- clamp-free variant, floors, address arithmetic, LDS staging

## 1. Clamp-free variant `band2_kernel<P, true>` (kept, default where proven)

The per-pixel tail carried four clamps that can never bind on standard-dynamic-range content: `L_bkg = max(., 0.1)`
(`fvvdp_lpyr_dec.py:265`), `contrast <= 1000` (`:266`, one `v_min` per plane), the clamp of the CSF query to the table's Y axis
(`fvvdp.py:530`) and the clamp of the interval index.  The library PROVES it per call from the display model's output range, the
RGB->Y weights and the filter taps (`luminance_range` / `clamps_never_bind` in `csrc/fvvdp_hip.hip`; standard_4k: luminances in
[0.598, 200], the contrast clamp would need a difference of 598 where no plane spans more than 253) and launches the variant without
them: loop body of two stages (20 band pixels per lane, `tools/isa_loop_hist.py`) **1535 -> 1291 VALU instructions** (v_min 160 -> 80,
v_med3 40 -> 0, v_max 46 -> 6, v_sub 160 -> 80 + packed adds).  Bit-identical results
(`tests/test_gpu_fused.py::test_clamp_free_variant_is_taken_only_where_proven_and_changes_nothing`); HDR displays with a black level
below 0.1 cd/m^2 and sources that hand over their own luminance take the variant with the clamps.

Same-box A/B (session 6, `FVVDP_BAND_INRANGE=0` = with clamps):

%s

## 2. Floors after that change (session 7, one box; ablation builds `-DBAND2_ABLATE=1` = no per-pixel tail, `-DBAND2_ABLATE_MEM` = rows from L2, no store)

```
%s
```

Clamp-free: arithmetic alone 28.7-28.9 us, data flow alone 32.2-32.3 us, together 35.1-35.6 us on that box (a slow one: the same build
measures 31.3-31.9 us in session 6 and 33.0-33.1 us in session 8).  With the clamps the arithmetic alone was 31.1-31.8 us.  The data
flow (153.9 MB per frame at 4.8-5.1 TB/s) is now the longer of the two floors; the kernel sits 9 %% above it.

## 3. Rows through a buffer resource, `v_cvt_flr_i32_f32` / `v_fract_f32` for the table index (kept)

Scalar row offset + loop-invariant lane offset instead of one 64-bit `v_lshl_add_u64` per load; 1291 -> 1275 VALU instructions per loop
body.  Same-box A/B (session 8; `previous` = the build before the change), pyramid levels us per frame, and the foveated clip of
configs[3] (level 0: 41.5 -> 40.8 us):

```
%s
== foveated, new
%s
== foveated, previous
%s
```

## 4. LDS-staged row prefetch two steps ahead (VERDICT r3 item 4) -- built, parity green, NOT faster, off (`-DBAND2_LDS_STAGE=1`)

`buffer_load_dwordx4 ... lds` (LDS-DMA, no destination registers) for the rows of step c+2, `s_waitcnt vmcnt(4)` + four `ds_read_b128`
for the rows of step c+1 at the top of step c, 12 KB of LDS per wave (12 waves per CU = 156 of 160 KB with the CSF tables), the same
register ring, 157 registers, 3 waves per SIMD.  62 parity tests pass on that build (`FVVDP_LIB=build_variants/r4_stage.so`), the
pooled sums are bit-identical.  Same-box A/B (session 10):

```
%s
```

3-5 %% slower in the pass-only loop, within 1 %% in the bench line.  A request already has a whole step (~2.4 us) to come back and a
fourth wave per SIMD did not lower the data-flow floor either (round 3): the kernel is not waiting for latency, and the staging
costs 16 LDS reads and ~10 vector instructions per loop body.
""" % (table(lines("r4s6", "*.json")), rd("r4s7/floors.txt"), rd("r4s8/ab.txt"), rd("r4s8/fov_new.txt"), rd("r4s8/fov_prev.txt"), rd("r4s10/ab.txt")))

# ---------------------------------------------------------------------------------------------------------------------------
write("r04_lockstep.md", """# Round 4: occupancy over ONE launch of the dominant kernel, and adjacent strips walked in step

## 1. Timeline of a launch (`tools/gpu_timeline.py`, profiling build `-DBAND2_TIMELINE`: every workgroup records start / end on the 100 MHz wall clock, XCD and CU)

`band2_kernel<4, true>`, 3840x2160 x60 (15120 single-wave work items, 3072 resident = 3 per SIMD) and 1920x1080 x60, session 19:

```
%s
%s
```

What it shows: (i) the wave slots are full until the last item starts; the last ~390 us (one item) run at falling occupancy: 12 %% of the
wave-time of the launch is idle, all of it there (4.92 "rounds" of 3072); (ii) the XCDs finish within 8 %% of each other; (iii) **the 3072
items of the first round, which start together and stay in step, take 316 us, every later item 380-415 us** -- the same work, the same
occupancy.  Free-running neighbours drift apart; in step they meet in the caches on the 20 of 128 columns they share and ask DRAM for one
contiguous piece of each row.  At 1920x1080 the whole launch (2.8 rounds of short items) stays in step by itself.

## 2. Adjacent strips in one workgroup, one barrier per stage (`FVVDP_BAND2_WPB`)

Workgroup = w waves = w adjacent strips of one chunk and frame, `s_barrier` at the top of every stage (4 level-A rows).  Same work items,
same partial sums, bit-identical results (`tests/test_gpu_fused.py::test_waves_per_workgroup_*`).  ms per 4K x 60 pair, K1 / levels 0+1 / all
levels us per frame (HIP-event medians), one gpurun call = one box per block:

Session 20 (compile-time w; 1920x1080 below):

```
%s
```

Timeline of the w = 4 launch on that box (steady items 374 us instead of 390; the first round still 314):

```
%s
```

Session 21 (6, 9 and 12 waves: a workgroup of 6 or 9 waves does not spread evenly over the 4 SIMDs; 12 = the whole CU, no gain):

```
%s
```

Session 22 (launch-time w; chunk heights; other frame sizes and batch lengths):

```
%s
```

Session 24 (a box in the fast K1 mode; 8 waves; a second barrier per stage; widths between 2048 and 3200):

```
%s
```

Session 25 (which wave takes which strip rotated with the work item -- no help for partly filled last groups: a workgroup holds its four
wave slots, so 30 strips in groups of 4 cost the launch the slots of 32):

```
%s
```

Shipped rule: 4 waves per workgroup where level A has >= 2560 columns AND the strips fill whole groups (4K: 36 strips, 8K: 72,
2560x1440: 24), else 1.  4K: -1.5 ... -6 %% on five boxes; 8K -2.6 %%; 2560x1440 0 ... -3 %%; 1920x1080 and 3200x1800 would lose 5-9 %%.

## 3. Short chunks dispatched last (`FVVDP_BAND2_KR2`, session 26) -- kept

The last tall chunk of every frame is cut into four (kr2 = kr / 4 level-C rows) and the short chunks of ALL frames get the highest
block indices, so they are dispatched after every tall one.  `kr2=0` = uniform chunks; the other values are explicit heights:

```
%s
```

Timeline with the short chunks (mean residency 2688 -> 2840 of 3072 waves; what is left of the tail is the XCDs finishing 5-7 %% apart):

```
%s
```

Also since session 27: levels of >= 0.5 Mpixel take the two-level kernel (was 1.5): at 4K levels 2+3 in one launch, 3.06 -> 2.67 us per frame:

```
%s
```

## 4. Resident workgroups with per-XCD work queues (session 28) -- built, 3.5 %% slower, removed

Grid = resident capacity; every workgroup takes item numbers from the atomic queue of the XCD it runs on (`s_getreg XCC_ID`), the next
number is requested before the item and looked at after it, an empty queue steals from the others.  The stealing works (2629-2825 items per
XCD instead of 2700 each), but the loop around the item costs the kernel its register allocation (13-18 scalar spills, 167-168 vector
registers) and the launch is slower than the hardware's dispatch of one workgroup per item, in the same build and against the previous one:

```
%s
```

## 5. The temporal kernel: timeline and pixel blocks from a ticket counter (sessions 31-33) -- kept for uint8 with 9..16 taps

`temporal_vec_kernel<8, 4, 0, 1>`, 3840x2160 x60 (32400 single-wave workgroups, 4096 resident), profiling build `-DK1_TIMELINE`:

```
%s
```

10 %% of the wave-time idle: the last item's length (items that start when the slots empty run faster -- the kernel is bound by memory),
and two of the eight XCDs (1 and 5 on that box) finish 6 %% after the others although the hardware hands every XCD exactly 4050 workgroups.
With a ticket counter (`TemporalArgs::ticket`) the grid is the resident capacity and a workgroup that has finished a block of pixels takes
the next one; the blocks have no neighbours to share anything with, so any order does.  Same blocks, same arithmetic
(`test_ticketed_temporal_kernel_changes_no_bits`).  `ticket=0` = one workgroup per block; ms per pair, K1 us per frame; one box per block:

```
%s
# second box
%s
# third box
%s
```

30 fps (8-slot ring): -3 %% ... +1 %%; 60 fps (16-slot ring, 2 pixels per lane, twice the blocks): -4 ... -6 %% on all three boxes.
Two more boxes (session 37, which also tried a STATIC weighting: every k-th workgroup that lands on XCD 1 or 5 idles, `skip=k` -- -2 %% at
best, +4 %% for one k on one box: dropped): tickets +0.5 / +1.5 %% with the 8-slot ring, -5.5 / -1.7 %% with the 16-slot ring.  Shipped:
tickets for 9..16 taps (33-64 fps) only.

```
%s
# second box
%s
```

Per-block timelines of both modes on a fourth box (session 35; the profiling build records every block inside the ticket loop).  Without
tickets XCDs 1 and 5 are again the ones that finish 110 us after the rest (the same two as on the first box: structural, not a box's
quirk).  With tickets every XCD ends within 55 us of the others and takes 3918-4185 blocks instead of 4050 each -- but the blocks, all
started within 35 us and never re-placed by the hardware, run in bursts (durations of 120 / 180 / 360 us where the hardware's dispatch
gives a steady 205), the wave-time per block is 7 %% higher, and the launch is not shorter at 30 fps on this box (1914 vs 1897 us):

```
%s
%s
```

## 6. A soft barrier across ALL strip groups of a (frame, chunk) row (session 39) -- far slower, removed

If items in step are 20 %% faster, why stop at 4 strips?  Wider workgroups do not fit the CU's wave slots (section 2), so the nine
workgroups of a 4K row were made to wait for each other every N stages through a counter in L2 (atomic add, bounded polling; nothing but
the timing depends on it).  Every use costs about 6 us -- the nine workgroups sit on different CUs with different company and the row runs
at the pace of its slowest member -- and the launch is slower at every interval (levels 0+1, us per frame; `sync_every=0` = off):

```
%s
```

Session 42, the other way of keeping the whole chip in step without communication: a metronome -- every wave starts its stages on
multiples of T ticks of the 100 MHz wall clock (`T=0` = off; a stage takes about 450 ticks free-running, 380 in the first round).
Slower at every T, and with it even the first-round items take 348 us instead of 314: whatever makes the first round fast, it is not
that the waves are in phase.

```
%s
```

## 7. The foveated one-level kernel in step (`-DFOV_LOCKSTEP`, session 23) -- no effect, removed

`band_kernel<4, false, 1>` already runs 4 waves per workgroup (4 frames of one tile).  Variant: the 4 waves take adjacent strips of one
chunk and frame (frame fastest over workgroups, so the rho-map slice stays in L2), one or two barriers per 4 coarse rows.  Parity green
(13 foveated tests); `tools/gpu_config4.py`, kernel us per frame [K1, level 0, level 1, ...]:

```
%s
```

Level 0: 37.7 / 37.7 (base) vs 38.4 / 37.3 (one barrier) vs 38.2 / 37.1 (two): the kernel is bound by its arithmetic, not by where its rows
come from.  The code was removed again.
""" % (rd("r4s19/timeline_4k.txt"), rd("r4s19/timeline_fhd.txt"), rd("r4s20/ab.txt"), rd("r4s20/timeline_4k_wpb4.txt"), rd("r4s21/ab.txt"),
       rd("r4s22/scan.txt"), rd("r4s24/scan.txt"), rd("r4s25/scan.txt"),
       rd("r4s26/scan.txt"), rd("r4s26/timeline_4k.txt"), rd("r4s27/scan.txt"), rd("r4s28/scan.txt"),
       rd("r4s31/k1_timeline.txt"), rd("r4s32/scan.txt"), rd("r4s33/scan_box1.txt"), rd("r4s33/scan_box2.txt"),
       rd("r4s37/scan_box1.txt"), rd("r4s37/scan_box2.txt"),
       "# FVVDP_K1_TICKET=1\n" + rd("r4s35/k1_timeline_ticket1.txt"), "# FVVDP_K1_TICKET=0\n" + rd("r4s35/k1_timeline_ticket0.txt"),
       rd("r4s39/scan.txt"), rd("r4s42/scan.txt"),
       "\n".join(l for l in rd("r4s23/fov.txt").split("\n") if l.startswith("==") or l.startswith("kernel us/frame:"))))

# ---------------------------------------------------------------------------------------------------------------------------
write("r04_yuv_counters.md", """# Round 4: what bounds the 10-bit 4:2:0 YUV temporal kernel at 60 fps (VERDICT r3 weak 9)

`temporal_yuv_vec_kernel<16, unsigned short, true, sRGB>` (3840x2160 x60 at 60 fps: 15 taps, 74 source frames per launch) takes 63-65 us per
output frame = 2.8-2.9 TB/s on its 182.5 MB per frame (0.36 of the peak).  SQ counters of the kernel, two `rocprofv3 --pmc` passes
(`tools/experiments/r4_session30.sh`, `tools/pmc_sq_summary.py`):

%s

Reading: two waves per SIMD are resident (`__launch_bounds__(64, 2)`, 128 vector registers) and a wave issues vector instructions during
0.436 of its resident time -- the vector ALU is busy 0.87 of the time, the waves wait for memory 0.22 of theirs, LDS and scalar work are
noise.  1.575e9 vector instructions / 33 450 waves / 74 frames = **636 per wave and source frame** (4 pixels x 2 streams per lane), of which
48 are transcendental (quarter rate).  Where they go, from the kernel source (`csrc/temporal_kernels.hpp`, `temporal_yuv_vec_body`):
unpack + convert 10-bit samples ~64; chroma bilinear x2 ~40; YCbCr matrix ~36; clamp to [0, 1] 32; display model (sRGB: compare, log2, multiply,
exp2, fused multiply-add, select per value, 24 values as 12 packed pairs) ~170; luminance 12; the two 15-tap filters on packed (test,
reference) pairs 120; ring bookkeeping ~30; out-of-range flag, stores ~30.  At 4 cycles per instruction and 16 per transcendental that is
~3100 cycles per wave and frame = 42 us per frame with the ALU never idle -- against 23 us for the bytes.  The reference evaluates the same
expressions per pixel (float RGB from the upsampled chroma, then the display model: `video_source_file.py:219-276`,
`fvvdp_display_model.py:147-165`), so a table cannot replace the display model here as it does for 8-bit RGB sources.  Of the 63-65 us,
14 of 74 processed frames (19 %%) are the filter's warm-up at 60 fps; a 120-frame clip pays 11 %%.

Left as it is: the kernel is bound by arithmetic that the algorithm prescribes; 2 pixels per lane instead of 4 would not change the count
per pixel.
""" % rd("r4s30/pmc_sq_yuv.md"))
