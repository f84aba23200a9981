#!/usr/bin/env python3
"""Assembles profiles/r05_k1_mode.md from the logs of tools/experiments/r5_session1..7.sh (gpurun_out/r5s*/): the temporal kernel's
slow / fast destination buffers -- microbenchmark, counters, and what the physical memory has to do with it."""
import os
import re

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")


def rd(rel, maxlines=None, cut=None):
    try:
        lines = open(os.path.join(G, rel)).read().rstrip("\n").split("\n")
    except OSError:
        return "(log %s not available)" % rel
    if maxlines:
        lines = lines[:maxlines]
    if cut:
        lines = [l[:cut] for l in lines]
    return "\n".join(lines)


def first_table(rel, ncols):
    """first round of the stream table, the first `ncols` variant columns"""
    out = []
    for l in rd(rel).split("\n"):
        if l.startswith("-- second round"):
            break
        p = l.split()
        if l.startswith("buf"):
            out.append(" ".join("%8s" % x for x in p[:3 + ncols]))
        elif p and p[0].isdigit():
            out.append(" ".join("%8s" % x for x in p[:3 + ncols]))
    return "\n".join(out)


def corr_lines(rel):
    return "\n".join(l for l in rd(rel).split("\n") if l.startswith("corr("))


def pmc_rows(rel, counters):
    """per pass of pmc_k1.md: the correlation lines of the named counters"""
    txt = rd(rel)
    out = []
    for blk in txt.split("### pass")[1:]:
        spread = re.search(r"spread of us/frame: ([\d.]+) \.\. ([\d.]+)", blk)
        for l in blk.split("\n"):
            m = re.match(r"\s+(\S+)\s+sum ([+-][\d.]+)", l)
            if m and m.group(1) in counters:
                out.append("| %s | %s | %s .. %s |" % (m.group(1), m.group(2), spread.group(1), spread.group(2)))
    return "\n".join(out)


def pmc_table(rel, counter_pass):
    txt = rd(rel)
    for blk in txt.split("### pass")[1:]:
        if counter_pass in blk.split("\n")[0]:
            rows = [l for l in blk.split("\n") if l.startswith("|")]
            # keep buf, kind, us/frame and the sum column of every counter
            hdr = [c.strip() for c in rows[0].strip("|").split("|")]
            keep = [i for i, h in enumerate(hdr) if h in ("buf", "kind", "us/frame") or h.endswith(" sum")]
            out = []
            for r in rows:
                cells = [c.strip() for c in r.strip("|").split("|")]
                if set(cells[0]) <= set("-"):
                    out.append("|" + "---|" * len(keep))
                    continue
                out.append("| " + " | ".join(cells[i] for i in keep) + " |")
            return "\n".join(out)
    return "(pass not found)"


doc = """# Round 5 -- why the temporal kernel (K1) is 20 %% slower on some level-0 buffers, and what can be done about it

K1 = `temporal_vec_kernel<8,4,u8>`: 6 B read + 16 B written per pixel and frame, 4K x 60: 31-33 us per frame on some destination
buffers, 36-38 on others (VERDICT r4 item 1; rounds 2-4 called them placement modes).  Everything below comes from ONE standalone
program, `tools/microbench/k1_stream.hip` (it includes the library's kernel from `csrc/`), run through `tools/experiments/r5_session1..7.sh`;
this file is assembled from their logs by `tools/make_r5_evidence.py`.  All sessions ran on the same MI355X box within one hour.

**Findings.**
1. *The mode is not in the kernel.*  A kernel that only replays K1's address stream (6 dword loads one frame ahead, four 1 KiB
   `float4` store runs per wave and frame, no table, no filter) takes the same time as K1 on every one of 13 buffers held at once
   (correlation 0.997 / 0.986 across the buffers); the store stream ALONE separates them (19.0 against 24.5 us), the load stream alone
   does not (5.66 us on every buffer); so does a plain grid-stride streaming write of the same 7.96 GB (7.0 against 5.4-5.6 TB/s).
   Block order (XCD-contiguous, runs of 16), store cache policy, occupancy (1 / 2 / 4 / 8 waves per SIMD), filler arithmetic,
   launching the frame in 4 or 8 slabs: the gap stays.  Section 1.
2. *One counter separates the modes:* `TCC_EA0_WRREQ_DRAM_CREDIT_STALL` (= `TCC_EA0_WRREQ_STALL`): 1.5-5.4 M cycles per launch on the
   fast buffers, 17-32 M on the slow ones (r = +0.976 with the duration over the 13 buffers of the pass), with `TCC_TAG_STALL` (0.3 M
   against 13 M, r = +0.994) behind it: on a slow buffer the L2's write requests wait for DRAM credits.  The stalls are spread over
   all 128 L2 channels (max / mean 2-4, the odd XCDs about twice the even ones), not concentrated in one.  Translation counters
   (`TCP_UTCL1_*`) do not separate them; bytes moved are the same (FETCH_SIZE x2 + WRITE_SIZE = 11.33 GB per launch: 3.37 GB read
   incl. the 7 history frames, 7.96 GB written = 1.035 x the algorithmic 10.95 GB).  Section 2.
3. *The cause is the physical memory behind the buffer.*  Physical chunks of 1 GiB, allocated and mapped one after the other, fall
   into TWO classes: writing one chunk alone runs at 5.1-5.8 TB/s; writing two chunks of the SAME class at once, the same; writing two
   chunks of DIFFERENT classes at once, 6.9-7.1 TB/s (8 x 8 pairwise table: consistent, transitive).  In allocation order the classes
   come in runs of 16-64 GiB (A x29, B x16, A x32, B x64, A x32, B x27 in session 4).  A level-0 buffer that lies in one class is the
   slow mode (every GiB of it writes at 5.2-5.5 TB/s and so does the whole); one whose pieces come from both classes is the fast mode
   (every GiB alone still 5.3, the whole 7.0) -- whatever the allocation API and the chunk size.  Reads depend on it by 2-3 %% only.
   Section 3.
4. *Which class an allocation gets is the driver's state, not the caller's choice.*  `hipMalloc`, 2 MB / 32 MB / 64 MB / 256 MB /
   1 GiB chunks: each was all-fast in one process and all-slow in another on the same box (sessions 1, 2, 6, 7); within one process the
   candidates tend to be alike (12 of 13 slow in session 2, 0 of 50 fast in five consecutive processes of session 7, then 20 of 20
   fast after a process that had allocated and freed 150 GiB); holding back a spacer between two halves of a buffer does not move
   the second half (`hipMemCreate` alone does not seem to place memory).  A search for the other class by allocating ahead found it
   after 29 GiB in one run and not within 100-120 GiB in three others.  Section 4.
5. *What the library does with that.*  (a) Level 0 of a large video context lives in TWO ranges -- even frame slots in one, odd slots in
   the other -- that `fvvdp_ctx_create` chooses among N = 6 half-size candidates (chunk-mapped and `hipMalloc` in turn) by writing every
   pair at once: the pair with the highest rate is two ranges of different classes wherever the candidates hold both (58 of 58
   single-context processes on five boxes: pair kept 6.9-7.2 TB/s, slowest pair 5.3-6.5, K1 29.7-31.5 us per frame); where the six lie
   in one class -- 2 of 8 contexts held in one process in session 27 -- up to 8 further candidates are taken while the first are held,
   until a pair of different classes turns up (session 28: K1 29.7-31.3 us on 24 of 24 contexts of three processes) -- section 7,
   which also lists the one session whose pair rates were not recorded and that had two processes at 33 and 35 us.  (Before that, the context kept the best of six full-size candidates, which helps only when a whole 8 GB candidate
   happens to span both classes: sections 5 and 6.)  0.1 s at creation; nothing happens in per-frame calls any more.  (b) K1 itself got what the replay showed
   to help on EVERY buffer: 4 waves per workgroup on adjacent pixel blocks (`k1_4w` / `rp_4w` columns: -1.2 ... -1.5 us per frame, slow
   and fast alike).  A kernel-side remedy for the slow mode does not exist: a plain streaming write has it.

## 1. One process, 13 level-0 candidates, the real kernel and replays of its stream (session 1; us per 4K frame)

`k1` the library's kernel, `rp` the replay, `k1_xcd` / `k1_x16` / `rp_xcd` / `rp_x16` other block orders, `rp_alu` with 96 filler
multiply-adds per step, `rp_w` stores only, `rp_r` loads only, `rp_4w` four waves per workgroup, `rp_4wx` that + XCD-contiguous,
`rp_occ8` 8 waves per SIMD, `rp_st0` stores without `nt`.

```
%s
```

```
%s
```

Session 2 (same box ten minutes later, more variants; `k1_2w/4w/8w` the real kernel with 2 / 4 / 8 waves per workgroup, `rp_4ws...` with a
barrier per frame, `rp_occ2/1` 2 / 1 waves per SIMD, `rp_slab8/4` the frame in 8 / 4 launches, `rp_1st*` one round of resident waves
scaled to the frame, `wr_seq` / `rd_seq` plain streaming write / read of the buffer).  This time 12 of the 13 buffers are slow:

```
%s
```

```
%s
```

## 2. Counters per buffer (session 1: one `rocprofv3 --pmc` pass per group, every pass its own process = its own draw)

The pass with the write-request counters (sum over the 128 L2 channels, per launch of 60 frames):

%s

Correlation of every counter collected with the duration, over the 13 buffers of its pass:

| counter | r with us/frame | spread of us/frame in that pass |
|---|---|---|
%s

Per-instance values of two passes (JSON output keeps the channel dimension; `tools/pmc_instances.py`, session 2): the stall cycles
per XCD (16 channels each), first buffers of the pass:

```
%s
```

## 3. Two classes of physical memory (session 4: 200 chunks of 1 GiB in allocation order; session 3: every GiB of 13 buffers)

Streaming-write rate [TB/s] of chunk i alone, together with chunk 0, with its successor, with chunk 100:

```
%s
```
(... 200 rows; classes along the allocation order, A = as chunk 0:)
```
%s
```

Pairwise, eight chunks spread over the 200:

```
%s
```

Every GiB of every candidate buffer, the whole buffer, and K1 on it (session 3):

```
%s
```

## 4. The allocator decides (sessions 5-7)

Level-0 candidates assembled from classified chunks, chunk sizes 1 GiB / 256 MB / 64 MB / 1 GiB in four consecutive processes (session 6;
with 64-256 MB chunks every 1 GiB group was already a mixture, every layout fast; with 1 GiB chunks 100 GiB in a row were one class):

```
%s
```

Two halves allocated S GiB apart (spacer allocated between them, released afterwards), interleaved chunk by chunk (session 7):

```
%s
```

## 5. Other kinds of device memory as candidates (sessions 11-13, a second box)

`fine` = `hipExtMallocWithFlags(hipDeviceMallocFinegrained)`, `uncached` = `hipDeviceMallocUncached`; columns: the real kernel (4 waves per
workgroup), the store-stream replay, plain streaming write and read of the buffer, us per 4K frame.  Uncached memory takes streaming
writes faster than cached memory of the same placement (18.2-18.4 us = 7.2-7.3 TB/s where the best cached buffers reach 18.8-19.0;
20-23 where a cached buffer in one class needs 24) and reads as fast; but it is a draw like the others (7 of 9 fast in session 13, where
`hipMalloc` had 7 of 9 and the chunk mappings 0 of 9 -- the reverse of session 1):

```
%s
```

The library with one kind forced (no choice) and with the choice among the kinds, alternating processes (`bench.py`; session 12; the rows
`uncached` of this session ran chunk-mapped -- the switch did not reach one-candidate contexts yet --; `best-of4` = chunk-mapped, hipMalloc,
uncached, chunk-mapped), then uncached only / best of four uncached (session 13):

```
%s
%s
```

The default since then: six candidates, two of each kind.

## 6. Both classes ARE within reach of one process -- and a level 0 assembled from them on purpose (sessions 15, 16, 19)

Pairwise: first halves of buffers i and j (12 candidates of 8 GB, four kinds, one process) written at once, TB/s; diagonal = buffer i alone.
The buffers split into two groups in every process; any two of different groups reach 6.9-7.3 TB/s together (session 15, process 1 of 3):

```
%s
```

Built on that (`tools/experiments/patches/r05_balance_levels_across_memory_classes.patch`): at context creation, probe the write rate of
level 0; if it lies in one class, allocate further chunk mappings behind other allocations until one writes fast TOGETHER with it, map the
32 MB chunks of the two alternately into level 0's range (no kernel sees a difference), rebuild levels 1-2 from the chunks left over.
Session 16, three processes in a row: a mapping of the other class after ONE further allocation every time, whole-buffer write rate 5.6-6.0
-> 6.5-6.6 TB/s, K1 30.9-31.3 us, creation 66-83 ms (best of six: 118-128 ms, K1 31.0-33.0):

```
%s
```

Session 19, the same box an hour later: every chunk mapping within four tries was dominated by the SAME class as the first (level 0 as
allocated 5.8-6.2 TB/s, no pair above 1.1 x that), the search fell through to the best of six candidates (which found `hipMalloc` buffers
at 68 us as it does without the search) -- after 64-128 GB of allocations whose release made the following ones wait: 4.5 s for the
first call.  Chunk mappings alone do not reach both classes reliably; the candidates of three kinds do more often, at a bounded price.
The search is not in the product.

```
%s
```

## 7. Level 0 in two chosen ranges (sessions 20-25): the fast mode by construction

What section 6 asked for, without remapping anything: frame slot s of level 0 lies at `(s & 1 ? hi : lo) + (s >> 1) * frame` (`L0Addr`,
`csrc/device_common.hpp`; one address formula in the temporal kernels, both pyramid kernels, the planar hand-over / export and the heat-map
colouring -- 186 GPU tests green, and the whole suite once more with EVERY context split, `FVVDP_LEVEL0_SPLIT=1`: 180 green).  At creation
a context whose level 0 holds >= 1 GiB gives back the range it came with, takes N = 6 candidates of HALF a level 0 (chunk-mapped and
`hipMalloc` in turn), writes every one of the 15 pairs at once with a streaming-write kernel (4 ms per pair: even workgroups into one
range, odd ones into the other), keeps the pair with the highest rate and frees the rest: ~90 ms, N halves of level 0 held for the
moment, no allocation / free / synchronisation in any per-frame call (`fvvdp_ctx_call_stats`, `test_level0_in_two_chosen_ranges_changes_no_bits`).
Consecutive frames of the temporal kernel's store stream then go to different classes of memory whenever the candidates hold both.

Ten pairs of bench processes on one box, the choice on / off (`FVVDP_PLACEMENT_PROBE=0`: one range as allocated); columns: ms per step,
K1 us per frame, levels 0+1, all levels | candidates kept, TB/s of the pair kept / of the slowest pair, K1 + pyramid pass at creation
(variant with clamps, first touch included), first step incl. creation [ms], JOD:

```
%s
```

Twelve processes in a row on another box (K1 of each of the 12 timed launches):

```
%s
```

And after six 1080p processes on a third (1080p: level 0 of 60 frames holds 2 GB, chosen the same way):

```
%s
```

Twenty more in a row on a fourth box with the card's sensors read before each (junction 46 C, memory 36-37 C, mclk 2000 MHz throughout:
no drift with time or temperature):

```
%s
```

K1 over the 58 two-range 4K processes of sessions 20, 22, 24, 25 and 26: 29.7-31.5 us per frame; one range as allocated on the same boxes:
30.6-32.5 (these boxes gave fast single ranges; the processes of sections 2, 4 and 5 that gave 35-37 are what the pairs are for -- in every
process so far the slowest pair of the six candidates ran at 5.3-6.5 TB/s, i.e. both classes were among them).  The exception, for the
record: session 23 (a fifth box; 1080p processes first, then four 4K ones whose `level0_alloc` was not printed) measured K1 at 29.8, 30.9,
33.1 and 35.1 us in that order, levels 0+1 at 30.0-30.7 throughout.  Session 25 repeated the sequence with the pair rates printed and
session 26 watched the sensors: neither reproduced it.  Whether those two contexts had no pair of different classes to choose from, or
something else slowed the writes, is not known (one candidate: the episodes of a few tens of milliseconds in which the power controller
pulls the clocks back, `profiles/r05_power.md`).

**Eight contexts held in one process** (`tools/experiments/gpu_alloc_draws.py 8`, the set-up VERDICT r4 item 1 names; K1 / levels 0+1 us per
frame per context, two rounds; the first line gives each context's layout code and the rate of the pair kept / of the slowest pair).
Session 27, before the further candidates existed: in the first process the six candidates of the 2nd and the 4th context lay in ONE class
(best pair 5.73 / 5.70 TB/s) and K1 ran at 35.5 / 34.6 us there; one range as allocated: 2 of 8 (chunk-mapped) and 3-4 of 8 (`hipMalloc`)
contexts slow in every process:

```
%s
```

Hence the further candidates: while no pair reaches 6.75 TB/s (pairs of one class: 5.3-6.5, of two: 6.9-7.2), `fvvdp_ctx_create` takes up
to 8 more half-size candidates WHILE HOLDING the first six -- so that the allocator has to move on -- and writes each together with one
range of the best pair, until a pair of different classes turns up.  Session 28, three processes of eight contexts: one context of 24
needed them (`+ 8 further ones`: 6.83 TB/s), K1 29.7-31.3 us on all 24 contexts in both rounds:

```
%s
```

**Uncached device memory is out.**  Round 5 first had `hipExtMallocWithFlags(hipDeviceMallocUncached)` among the candidate kinds
(5-15 %% faster streaming writes, section 5).  With every context split and the odd slots in a fresh uncached range, 3 of 180 GPU tests
failed and not always the same ones: the FIRST pass over a fresh range returned a few pixels that the temporal kernel had not written
(row 0 of a three-call test differed, rows 1-2 did not), never with `hipMalloc` memory in the same place (session 21: 3 / 0 / 1 / 0 failures
for uncached / hipMalloc / uncached after a fill / one range, twice).  The reads of a range that was cached under its previous owner are
not safe until its lines are gone; the kind was removed from the library rather than worked around.

```
%s
```

## 8. Reads and writes in different classes (session 31, `tools/microbench/rw_classes.hip`)

Eight ranges of 4 GB (chunk-mapped and `hipMalloc` in turn), classified by writing each together with range 0; then a kernel whose
workgroups either read range A or write range B as plain streams, the workgroups split in the ratio of the bytes: read only, 1 written per 16
read (the pyramid pass), 6 read per 16 written (the temporal kernel), 1:1.  TB/s of both streams together, two processes:

```
%s
```

(a) A write stream of 1/16 of the bytes costs a read stream 2-3 %% whichever classes the two ranges lie in: nothing to gain for the pyramid
pass from placing its small output elsewhere.  (b) The write-dominated mix depends on the DESTINATION alone (5.5-5.9 TB/s into a range of one
class, 6.1-6.9 into one that spans both -- ranges 2, 4 and 6 of the second process), not on where the source lies.  (c) Plain streams in
the temporal kernel's proportions reach 6.8-6.9 TB/s on such a destination = 26.5 us per 4K frame; the kernel, and the replay of its address
stream, take 29.7-31.5: the shape of the stream (six byte planes read 1 KB per workgroup and frame, sixty frames written 16 KB at a time
133 MB apart) costs ~12 %% against plain streaming, and none of the replay variants of section 1 recovers it.

## 9. What the temporal kernel pays for: reads MIXED into its writes (session 32, `k1_stream ... decoupled`)

The replay of the kernel's stream with its loads and stores in one wave (`rp`), the loads alone (`rp_r`), the stores alone (`rp_w`), and
the load-only and the store-only replay launched at the same time on two streams (`rp_r || rp_w`: same addresses, same instruction
shapes; the loads are done after a quarter of the time and the stores finish alone); last column: plain streams of the same sizes.
us per 4K frame, two processes:

```
%s
```

The stores alone run at the rate of a plain streaming write (19.3-19.5 us per frame on a range that spans both classes), the loads alone
take 5.6-5.8 us, one after the other 25.4-25.6 us -- and interleaved, as a kernel that turns every loaded frame into a stored one has them,
31.1-31.5 us: **the memory system charges ~6 us per frame (a fifth of the kernel) for the reads being mixed into the writes**, on slow and
fast ranges alike.  It is not the coupling inside a wave: with the load-only kernel throttled (fewer waves per CU through its LDS
footprint) so that its loads spread over the whole duration, the two kernels together take as long as the coupled replay again
(first pair of numbers per footprint: when the load kernel ended / when both had ended):

```
%s
```

So the gain needs PHASES -- the chip reading for a while, then writing for a while -- and a kernel can only get them by holding a phase's
worth of data on the chip (one 4K frame pair of sources is 50 MB) or by keeping all its waves in step.  Tried: every wave loads the bytes of
ten frames in one go (60 registers) and then stores ten frames, the loads gated into a window of the 100 MHz real-time counter that all waves
see (`replay_phased`, periods of 28-38 us, 20-28 %% of them open for loads, stores gated out of the window or not).  On a box whose four
ranges were all of one class (coupled replay 38.2 us per frame, loads then stores 31.3): ungated 37.8, gated 38.0-40.3 -- no phase gain;
the waves do not finish a block per period (the time does not follow the period), so the windows never line the chip up.  Also tried: the frame in 8 / 16 / 32 / 64 launches of one
round of resident waves or less, each wave first touching all source bytes of its block (a read-only phase at the start of every launch) so
that the loop would find them in the 256 MB memory-side cache: 41-45 us per frame against 33-34 without the touch (last block below) -- the
touch costs its 6-8 us and the loop gains nothing (a slab's sources, 104-416 MB, do not survive its 250-1000 MB of stores).  Not built
into the kernel; the 6 us stand as the difference between the temporal kernel (0.73-0.77 of the HBM peak) and plain streaming of its bytes.

```
%s
```

```
%s
```
"""

zones = rd("r5s4/zones.txt").split("\n")
cls = "".join(("B" if float(l.split()[2]) > 6.4 else "A") for l in zones[2:] if len(l.split()) == 5 and l.split()[0].isdigit())
pair = "\n".join(zones[zones.index([l for l in zones if l.startswith("pairwise")][0]):]) if any(l.startswith("pairwise") for l in zones) else ""
inst = "\n".join(l[:230] for l in rd("r5s2/instances0.md").split("\n") if l.startswith("| ") and ("DRAM_CREDIT" in l or "TAG_STALL" in l))[:6000]
out = doc % (first_table("r5s1/stream.txt", 13), corr_lines("r5s1/stream.txt"),
             first_table("r5s2/stream.txt", 29), corr_lines("r5s2/stream.txt"),
             pmc_table("r5s1/pmc_k1.md", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_LEVEL"),
             pmc_rows("r5s1/pmc_k1.md", {"TCC_EA0_WRREQ_DRAM_CREDIT_STALL", "TCC_EA0_WRREQ_STALL", "TCC_TAG_STALL", "TCC_TOO_MANY_EA_WRREQS_STALL",
                                         "TCC_EA0_WRREQ_LEVEL", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL", "TCC_EA0_RDREQ_LEVEL", "TCC_EA0_WRREQ", "TCC_EA0_RDREQ",
                                         "TCC_HIT", "TCC_MISS", "TCC_WRITEBACK", "TCP_UTCL1_TRANSLATION_MISS", "TCP_UTCL1_TRANSLATION_HIT",
                                         "TCP_TCC_READ_REQ_LATENCY", "TCP_TCC_WRITE_REQ_LATENCY", "TCP_PENDING_STALL_CYCLES", "TCC_BUSY",
                                         "GRBM_UTCL2_BUSY", "FETCH_SIZE", "WRITE_SIZE", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"}),
             inst,
             "\n".join(zones[:36]), cls, pair,
             rd("r5s3/regions.txt", 15),
             "\n".join(l for l in rd("r5s6_call.log").split("\n") if not l.startswith("[gpurun]")),
             rd("r5s7/spread.txt"),
             "\n".join(l for l in rd("r5s13_call.log").split("\n") if l.startswith("==") or (l[:1].isdigit() and len(l.split()) == 6) or l.startswith("buf kind")),
             rd("r5s12/kinds.txt"),
             "\n".join(l for l in rd("r5s13_call.log").split("\n") if l.startswith("uncached ")),
             "\n".join(rd("r5s15_call.log").split("\n")[5:19]),
             rd("r5s16/balance.txt", cut=330),
             "\n".join(l[:330] for l in rd("r5s19/ab.txt").split("\n") if l.startswith("4K") or l.startswith("fvvdp: level write")),
             rd("r5s22/ab.txt", cut=200), rd("r5s24/k1.txt", cut=260), rd("r5s25/out.txt", cut=260),
             "\n".join(l[:200] for l in rd("r5s26/drift.txt").split("\n") if l.startswith("run ")),
             rd("r5s27/draws.txt", cut=260), rd("r5s28/draws.txt", cut=260),
             "\n".join(l[:200] for l in rd("r5s21_call.log").split("\n") if l.startswith("==") or l.startswith("FAILED") or " passed" in l),
             rd("r5s31/rw.txt", cut=200),
             rd("r5s32/decoupled.txt", cut=200), "\n".join(l[:200] for l in rd("r5s32/throttled.txt").split("\n") if l[:1].isdigit() or l.startswith("buf")),
             "\n".join(l[:330] for l in rd("r5s32/phased.txt").split("\n") if "phased" in l or l[:1].isdigit()),
             "\n".join(l[:160] for l in rd("r5s32/touch.txt").split("\n") if "slabs" in l or l[:1].isdigit() or l.startswith("buf")))
open(os.path.join(R, "profiles", "r05_k1_mode.md"), "w").write(out)
print("written", len(out))
