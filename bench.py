#!/usr/bin/env python3
"""Benchmark of the FovVideoVDP hot path on MI355X.

A "step" is one full pass of the hot path over `--pairs-per-gpu` (default 1) synthetic 3840x2160 x 60-frame video pairs
per GPU (BASELINE.json configs[2]: uint8 RGB, standard_4k, 30 fps -> 8-tap temporal filter, foveated off; configs[4] is
`--gpus 8 --pairs-per-gpu 8`; `--config 1` / `--config 3` run configs[1] / configs[3]), inputs already resident in HBM: unpack + sRGB
display model + luminance + temporal filtering + pyramid + CSF + masking + pooling + JOD regression.  ONE step path for every N
(`step_path` in the line): the pairs of a step are queued back to back without host synchronisation (`predict_batch`; every pair's
pooling + JOD regression run in the library on the device), the per-pair result rows (Q_per_ch | range flag | JOD) of all ranks are
combined by ONE all-reduce of the device buffer and ONE device -> host copy ends the step.  Since round 6 the all-reduce is issued on ONE
rank too (`--collective auto`: a world-size-1 `nccl` group, RCCL initialised, `communicator` / `collective` in the line; `--collective
off` = the one-rank shortcut), so the N = 1 point of a scaling curve carries the collective's fixed cost.  `--shard frames` measures the
other decomposition of north_star: ONE pair of `--frames` x N frames whose output frames are split across the ranks
(`predict_frame_sharded`: every rank reads its own fl-1 frames of temporal halo, ONE all-reduce of Q_per_ch + range flag, pooling on every rank).
value = Mpixels/s (test+ref) = 2*W*H*frames of a step / median step seconds.

  python bench.py [--gpus N --steps K --warmup W --pairs-per-gpu P --shard pairs|frames --config 1|2|3 --collective auto|off|force]
  (N>1: launched by torch.distributed.run)

ONE JSON line on rank 0's stdout and nothing else (file descriptor 1 is pointed at stderr while the process runs: librccl prints a banner).
`roofline`: the dominant kernel (two-level pyramid kernel, levels 0+1), timed per launch with HIP events inside the library on the kernels'
stream (median/min/max over >= 10 launches); `roofline.frac_rocprof_avg`: the same bytes over the rocprofv3 AVERAGE duration of the kernel
in a kernel trace run from here (cold launches included); `roofline.frac_bmin`: SURVEY 8(d)'s conservative figure (level 0 read once, over the
whole graded pass); `roofline.with_clamps`: the variant of the kernel that HDR displays take; `graded_pass`: all pyramid levels + finalize
against SURVEY 8(d)'s 221.2 MB per 4K frame; `roofline.traffic`: HBM bytes per launch of that kernel from two rocprofv3 --pmc passes run from
here (fallback: the committed profile); `roofline_fov` (`--config 3`): the foveated pass; `cpu_baseline`: the numpy oracle on a bounded
sample of the same workload; `value_h2d_inclusive`: the same call on pageable host arrays (never `value`).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def level_sizes(W, H, n_bands):
    out = []
    w, h = W, H
    for _ in range(n_bands + 1):
        out.append((w, h))
        w, h = (w + 1) // 2, (h + 1) // 2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    # the context settles its level-0 ranges at creation, not over calls; what the warm-up is for: after an idle gap the clocks come up
    # over the first 6-8 steps (~30 ms; the per-launch list in profiles/r05_final_kernel_trace.md)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--shard", default="pairs", choices=["pairs", "frames"], help="pairs: independent pairs per rank (configs[4]); "
                    "frames: ONE pair of --frames x N frames, output frames split across the ranks (weak scaling in frames)")
    ap.add_argument("--pairs-per-gpu", type=int, default=1, help="independent pairs per rank and step (BASELINE configs[4]: 8)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--fps", type=int, default=30)
    ap.add_argument("--display", default="standard_4k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-array (PCIe-inclusive) timing")
    ap.add_argument("--timing-reps", type=int, default=12, help="launches of each kernel timed with HIP events")
    ap.add_argument("--no-measure-traffic", action="store_true", help="do not collect roofline.traffic live (two rocprofv3 --pmc "
                    "passes, FETCH_SIZE and WRITE_SIZE, of tools/gpu_bandonly.py in subprocesses, ~10 s); the committed "
                    "profiles/rNN_pmc_level0.json of the same kernel and launch shape is quoted instead")
    ap.add_argument("--measure-traffic", action="store_true", help=argparse.SUPPRESS)      # the default now
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the CPU baseline, one output frame each (0 = every core "
                    "the container may use (affinity mask and cgroup quota), up to 64, bounded by free memory at ~3 GB per 4K "
                    "worker; 1 = time the oracle in-process)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "dry-run the multi-rank code path on a single-GPU box)")
    ap.add_argument("--collective", default="auto", choices=["auto", "off", "force"],
                    help="one rank only (N > 1 always runs the collective).  force: a world-size-1 process group of --backend is created and "
                         "the result rows go through the SAME zero-buffer all-reduce as at N > 1 (RCCL initialised, ncclAllReduce "
                         "on the rows the library's kernels wrote), so the N = 1 point of a scaling curve carries the collective's fixed "
                         "cost and `communicator` reports the RCCL version; auto (default): force where the group initialises, "
                         "otherwise the one-rank shortcut with the reason in `communicator.error`; off: the shortcut")
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3],
                    help="BASELINE.json configs[] index: 1 = 1920x1080 x60 standard_fhd, 2 = 3840x2160 x60 standard_4k (the default "
                         "arguments), 3 = 3840x2160 x120 foveated with moving gaze on standard_hdr_pq (adds `roofline_fov`)")
    args = ap.parse_args()
    # stdout carries ONE JSON line and nothing else: librccl prints a version banner on the C stdout of rank 0 when a communicator is
    # created (5 lines, flushed at exit), other libraries may follow.  File descriptor 1 is pointed at stderr for the life of the
    # process and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    foveated = False
    if args.config == 1:
        args.width, args.height, args.frames, args.fps, args.display = 1920, 1080, 60, 30, "standard_fhd"
    elif args.config == 3:
        args.width, args.height, args.frames, args.fps, args.display = 3840, 2160, 120, 30, "standard_hdr_pq"
        foveated = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()      # dry run: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    comm_error = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)
    elif args.collective != "off":
        # ONE rank: the same communicator set-up and the same all-reduce as N > 1, on a world of one (in-process store: no port, no
        # rendezvous).  RCCL loads, ncclCommInitRank runs on this device, and every step's result rows pass through ncclAllReduce.
        import torch.distributed as dist
        try:
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", store=dist.HashStore(), rank=0, world_size=1, device_id=dev)
            else:
                dist.init_process_group(backend=args.backend, store=dist.HashStore(), rank=0, world_size=1)
            probe = torch.ones(4, device=dev)
            dist.all_reduce(probe)               # the communicator is created lazily on some builds: fail HERE, not in the timed region
            torch.cuda.synchronize(dev)
            assert float(probe.sum()) == 4.0
        except Exception as e:                   # auto: no communicator on this box -> the one-rank shortcut, with the reason in the line
            if args.collective == "force":
                raise
            comm_error = "%s: %s" % (type(e).__name__, str(e)[:300])
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
            dist = None
    assert args.gpus == world, "--gpus must equal the number of launched ranks"
    force_collective = dist is not None and world == 1
    comm = None
    if dist is not None:
        # what the communicator itself reports, gathered from every rank: the driver can check that N ranks on N devices met
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "device_name": torch.cuda.get_device_name(dev),
                "device_uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        ver = None
        if args.backend == "nccl":
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = None
        comm = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver, "ranks": allr,
                "distinct_devices": len(set((r["device_index"], r["device_uuid"]) for r in allr)),
                "collective_on_one_rank": "forced: world-size-1 group, the step's all-reduce is issued" if force_collective else None}
    elif comm_error is not None:
        comm = {"backend": None, "world_size": 1, "error": comm_error, "collective_on_one_rank": "shortcut (no communicator)"}

    import fovvideovdp_amd as fv
    from fovvideovdp_amd import _native as nat
    from fovvideovdp_amd.synth import synth_video_pair
    from fovvideovdp_amd.sharding import gather_pair_results

    W, H, N, fps, K = args.width, args.height, args.frames, args.fps, max(1, args.pairs_per_gpu)
    frames_mode = args.shard == "frames"
    if frames_mode:
        K = 1
    n_clip = N * world if frames_mode else N           # frames of one clip (frame sharding: --frames per rank, weak scaling)
    # ---- startup guard: what this rank is about to hold in HBM, against what is free (a clear error beats an OOM mid-run) ----
    es_in = 1                                          # uint8 RGB
    need = {"source clips": 2.0 * 3 * W * H * n_clip * es_in * K,
            "pyramid scratch (one batch of <= 128 frames, all levels)": 16.0 * W * H * 1.34 * min(N, 128),
            "six half-size level-0 candidates at context creation (two level 0 beyond the scratch)": 2 * 16.0 * W * H * min(N, 128),
            "margin": 2.0 * (1 << 30)}
    free_b, total_b = torch.cuda.mem_get_info(dev)
    if sum(need.values()) > free_b:
        raise SystemExit("bench.py rank %d: needs %.1f GB of HBM on %s (%s) but only %.1f of %.1f GB are free" % (
            rank, sum(need.values()) / 1e9, dev, ", ".join("%s %.1f GB" % (k, v / 1e9) for k, v in need.items()), free_b / 1e9, total_b / 1e9))
    # pair sharding: rank r owns pairs r*K .. r*K+K-1, generated on its own GPU (2 x 1.49 GB each at 4K x 60); frame sharding: every
    # rank holds the whole clip of N*world frames (it reads its own output frames + fl-1 frames of halo) and evaluates its share
    pairs = [synth_video_pair(n_clip, H, W, device=dev, pair=(0 if frames_mode else rank * K + k)) for k in range(K)]
    m = fv.fvvdp(display_name=args.display, device=dev, foveated=foveated)
    gaze = None
    if foveated:
        from fovvideovdp_amd.synth import synth_gaze
        gaze = synth_gaze(n_clip, H, W).numpy()        # moving gaze, top-left -> bottom-right (ex_foveated_video.py:36-37)
    step_path = ("predict_frame_sharded: this rank's output frames queued without host sync -> all-reduce of Q_per_ch -> pooling + JOD on every rank"
                 if frames_mode else
                 "predict_batch (pairs queued without host sync, pooling + JOD per pair in the library) -> all-reduce of the result rows "
                 "(%s) -> one device-to-host copy" % ("issued on this one rank too: world-size-1 %s group" % args.backend if force_collective
                                                      else ("one rank: skipped" if world == 1 else "%d ranks" % world)))

    jods_rowlen = []            # length of a result row (Q_per_ch | flag | JOD), noted by the first step
    pin = {}

    def rows_to_host(g):
        """the one device -> host copy of a step, through a page-locked staging buffer like `fvvdp._to_host` (an asynchronous copy + one
        stream synchronisation instead of a pageable `.cpu()`); falls back to `.cpu()` where page-locking is refused"""
        buf = pin.get("buf")
        if buf is False:
            return g.cpu()
        if buf is None or buf.shape != g.shape:
            try:
                buf = torch.empty(g.shape, dtype=g.dtype, pin_memory=True)
            except RuntimeError:
                buf = False
            pin["buf"] = buf
        if buf is False:
            return g.cpu()
        buf.copy_(g, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        return buf

    def step():
        if frames_mode:
            from fovvideovdp_amd.sharding import predict_frame_sharded
            vs = fv.fvvdp_video_source_array(pairs[0][0], pairs[0][1], fps, dim_order="BCFHW", display_photometry=m.display_photometry)
            q, _ = predict_frame_sharded(m, vs, rank, world, fixation_point=gaze, force_collective=force_collective)
            return [float(q)]
        # queue every pair without host synchronisation, then the one collective of the path (all-reduce of a zero buffer in which
        # this rank filled its own rows; one rank: nothing), ONE device -> host copy of [pairs, Q_per_ch | range flag | JOD]
        outs = m.predict_batch(pairs, dim_order="BCFHW", frames_per_second=fps, fixation_point=gaze)
        rows = outs[0][1]["result_buffer"].unsqueeze(0) if K == 1 else torch.stack([st["result_buffer"] for (_, st) in outs])
        if not jods_rowlen:
            jods_rowlen.append(rows.shape[1])
        return rows_to_host(gather_pair_results(rows, rank, world, force_collective=force_collective))[:, -1].tolist()

    def step_reference_call():
        q, _ = m.predict(pairs[0][0], pairs[0][1], dim_order="BCFHW", frames_per_second=fps, fixation_point=gaze)
        return [float(q)]                                            # the reference's call, incl. its host sync

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    t_first = time.perf_counter()
    for w in range(args.warmup):
        jods = step()
        if w == 0:
            torch.cuda.synchronize(dev)
            t_first = time.perf_counter() - t_first        # incl. context creation (scratch allocation + the level-0 choice)
    fence()
    t0 = time.perf_counter()
    stamps = [t0]
    for _ in range(args.steps):
        jods = step()
        stamps.append(time.perf_counter())       # every step ends in its own host synchronisation (the result copy)
    fence()
    dt = time.perf_counter() - t0
    # max over ranks of the timed region and of every single step (a step ends in the collective, so the ranks leave it together)
    tall = torch.tensor([dt] + [b - a for a, b in zip(stamps[:-1], stamps[1:])], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tall, op=dist.ReduceOp.MAX)
    tall = tall.cpu().numpy()
    dt = float(tall[0])
    step_s = np.sort(tall[1:])
    dt_med = float(np.median(step_s))
    px_step = 2.0 * W * H * (n_clip if frames_mode else N * K * world)
    # the collective alone (every rank takes part): the step's all-reduce on a buffer of the step's size, back to back, host clock
    # around a synchronised loop -> what one all-reduce adds to a step (launch + RCCL + the stream hand-over torch does around it)
    coll = None
    if dist is not None:
        nrow = int(jods_rowlen[0]) if jods_rowlen else 14 * n_clip + 1
        cbuf = torch.zeros((world * (1 if frames_mode else K), nrow), dtype=torch.float32, device=dev)
        for _ in range(5):
            dist.all_reduce(cbuf)
        torch.cuda.synchronize(dev)
        reps_c = 50
        t1 = time.perf_counter()
        for _ in range(reps_c):
            dist.all_reduce(cbuf)
        torch.cuda.synchronize(dev)
        tc_ = torch.tensor([(time.perf_counter() - t1) / reps_c], dtype=torch.float64, device=dev)
        dist.all_reduce(tc_, op=dist.ReduceOp.MAX)
        coll = {"op": "all_reduce(sum) of a zero-initialised fp32 buffer [%d rows x %d] = %d B in which every rank filled its own rows" % (
                    cbuf.shape[0], nrow, cbuf.numel() * 4),
                "us_per_call_back_to_back": round(float(tc_.cpu()[0]) * 1e6, 1), "calls_timed": reps_c, "backend": dist.get_backend(),
                "world_size": world}
    # the reference-style synchronous call next to the step path, once (one rank, one pair): the same kernels, the same single sync
    predict_ms = None
    if world == 1 and K == 1 and not frames_mode:
        # (the legs between the timed region and here leave the GPU idle for a few milliseconds; the launches after an idle gap run on clocks
        # that are still coming up -- profiles/r05_power.md -- and until round 6 this figure, taken on five such calls, read 0.4 ms above
        # the step although a back-to-back comparison shows none: tools/experiments/gpu_predict_overhead.py.  Warm up, then time.)
        for _ in range(8):
            step_reference_call()
        tp = []
        for _ in range(max(7, args.steps)):
            t1 = time.perf_counter()
            step_reference_call()
            tp.append(time.perf_counter() - t1)
        predict_ms = float(np.median(tp)) * 1e3
    mpix_mean = px_step * args.steps / dt / 1e6
    mpix = px_step / dt_med / 1e6                 # the metric is the MEDIAN of >= 10 runs after the warm-ups

    # ---- per-kernel timing: HIP events inside the library on the kernels' own stream, one reading per launch ----
    roof = None
    extra = {}
    test, ref = pairs[0]
    if rank == 0:
        # the per-kernel leg launches the same kernels on the same context (same HBM scratch, same placement) as the timed steps
        if frames_mode:
            test, ref = test[:, :, :N], ref[:, :, :N]
        gz = None if gaze is None else gaze[:N]
        m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps, fixation_point=gz)
        ctx = m._ctx
        st_, cm_, nt_, kept_ = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(-1)
        us_ = (C.c_float * 8)()
        cs_ = (C.c_int64 * 3)()
        nat.check(nat.lib().fvvdp_ctx_alloc_info(ctx.handle, C.byref(st_), C.byref(cm_), us_, 8, C.byref(nt_), C.byref(kept_)))
        nat.check(nat.lib().fvvdp_ctx_call_stats(ctx.handle, cs_))
        kinds = {0: "hipMalloc", 1: "virtual-memory API, physical chunks of %s MB" % os.environ.get("FVVDP_VMM_CHUNK_MB", "32")}
        two = cm_.value >= 100
        extra["level0_alloc"] = {
            # level 0 lives in two ranges (even / odd frame slots) that fvvdp_ctx_create chose among half-size candidates: the pair that a
            # streaming-write probe wrote fastest at once, i.e. two ranges of different classes of the box's physical memory where there are any
            "in_use": ("even frame slots: %s; odd frame slots: %s" % (kinds.get(cm_.value % 10), kinds.get((cm_.value // 10) % 10))) if two
                      else kinds.get(cm_.value, str(cm_.value)),
            "half_size_candidates": nt_.value, "further_candidates_tried": int(us_[3]), "kept_indices": [kept_.value % 8, kept_.value // 8] if kept_.value >= 0 else None,
            "pair_write_rate_tbs": {"kept": round(us_[1], 2), "lowest": round(us_[2], 2)} if nt_.value else None,
            "state": st_.value,
            "per_frame_calls": {"host_syncs": int(cs_[0]), "allocations": int(cs_[1]), "frees": int(cs_[2])},
            "first_step_ms_incl_context_creation": round(t_first * 1e3, 1) if args.warmup > 0 else None,
            "evidence": "profiles/r05_k1_mode.md"}
        nat.check(nat.lib().fvvdp_ctx_timing_enable(ctx.handle, 1))
        nk = 16 + 2
        ms = (C.c_float * nk)()
        cnt = (C.c_int32 * nk)()
        nat.check(nat.lib().fvvdp_ctx_timing_read(ctx.handle, ms, cnt, nk, 1))
        n_bands, batch = ctx.key[2], ctx.key[4]
        rows = []
        for _ in range(max(10, args.timing_reps)):
            m.predict(test, ref, dim_order="BCFHW", frames_per_second=fps, fixation_point=gz)
            nat.check(nat.lib().fvvdp_ctx_timing_read(ctx.handle, ms, cnt, nk, 1))
            rows.append([ms[i] for i in range(n_bands + 2)])          # ms per call: temporal, levels 0.., finalize
        nat.check(nat.lib().fvvdp_ctx_timing_enable(ctx.handle, 0))
        t = np.asarray(rows, dtype=np.float64) / N * 1e3              # us per frame
        sizes = level_sizes(W, H, n_bands)
        P = 4
        alg = [4.0 * P * (sizes[i][0] * sizes[i][1] + sizes[i + 1][0] * sizes[i + 1][1]) for i in range(n_bands)]   # bytes per frame
        # dominant kernel: the launch recorded at level 0.  A launch that covers two levels (band2_kernel) reports 0 for
        # the second one; its algorithmic bytes are SURVEY 8(d)'s per-level figure of BOTH levels (what a one-level-per-
        # pass pyramid streams), although level 1 never leaves the chip -- `traffic` is what it really moves.
        fused01 = n_bands > 1 and float(np.max(t[:, 2])) == 0.0
        b0 = alg[0] + (alg[1] if fused01 else 0.0)
        t0f = t[:, 1]
        med0 = float(np.median(t0f))
        ach = b0 / (med0 * 1e-6) / 1e9
        frames_per_launch = min(N, batch)
        roof = {"bound": "hbm",
                "kernel": "band2_kernel<4, true> pyramid levels 0+1 (reduce x2, expand x2, contrast, CSF, masking, pooling; <4, true> = the "
                          "variant without the clamps the library proved unreachable on this display, <4, false> otherwise)" if fused01
                          else ("band_kernel<4, false, 1> level 0, one level per launch (foveated: LUT slice + row table in LDS, per-pixel "
                                "eccentricity and resolution magnification, trilinear CSF query, masking, pooling)" if foveated
                                else "band_kernel<4> level 0 (pyramid+CSF+masking+pooling)"),
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": None, "avg_launch_ms": round(float(np.mean(t0f)) * frames_per_launch * 1e-3, 4),
                "median_launch_ms": round(med0 * frames_per_launch * 1e-3, 4),
                "min_launch_ms": round(float(np.min(t0f)) * frames_per_launch * 1e-3, 4),
                "max_launch_ms": round(float(np.max(t0f)) * frames_per_launch * 1e-3, 4), "launches_timed": len(rows),
                "bytes_per_launch": int(b0 * frames_per_launch), "frames_per_launch": frames_per_launch,
                "frac_at_min": round(b0 / (float(np.min(t0f)) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                # every reading, in launch order (ms per launch): whoever doubts the median can recompute it
                "launch_ms_all": [round(float(x) * frames_per_launch * 1e-3, 4) for x in t0f]}
        # HBM bytes per launch from the PMC counters cannot be collected from inside this process; they come from the
        # committed rocprofv3 --pmc passes of the same kernel and launch shape (tools/pmc_level0.py, profiles/)
        pdir = os.path.join(ROOT, "profiles")
        pmc = sorted(f for f in os.listdir(pdir) if f.endswith("pmc_level0.json")) if os.path.isdir(pdir) else []
        live = measure_traffic_live() if (world == 1 and not args.no_measure_traffic and
                                          (W, H, int(frames_per_launch)) == (3840, 2160, 60)) else None
        k1_live = live.get("k1") if live else None
        if live is not None and ("band2" in live.get("kernel", "")) == fused01:
            roof["traffic"] = int(live["traffic_bytes"])
            roof["traffic_source"] = "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gpu_bandonly.py (FETCH_SIZE x2 + WRITE_SIZE)"
        elif pmc and (W, H, int(frames_per_launch)) == (3840, 2160, 60):
            with open(os.path.join(pdir, pmc[-1])) as f:
                pj = json.load(f)
            if ("band2" in pj.get("kernel", "")) == fused01:
                roof["traffic"] = int(pj["traffic_bytes"])
                roof["traffic_source"] = "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)" % pmc[-1]
        if live is not None and live.get("trace") and ("band2" in live.get("kernel", "")) == fused01:
            tr = live["trace"]
            roof["rocprof_avg_launch_ms"] = round(tr["avg_us"] * 1e-3, 4)
            roof["frac_rocprof_avg"] = round(roof["bytes_per_launch"] / (tr["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            roof["rocprof_note"] = ("rocprofv3 --kernel-trace of tools/gpu_bandonly.py (whole predict() calls on the same workload) run from "
                                    "here: AVERAGE over all %d dispatches of this kernel, the cold-clock launches at the start included "
                                    "(median %.4f ms, steady median after the first 3: %.4f ms) -- the conservative reading; `frac` is the "
                                    "HIP-event median of this process" % (tr["n"], tr["median_us"] * 1e-3, tr["steady_median_us"] * 1e-3))
        if roof["traffic"] is not None:
            # `achieved` credits the level-1 write + read-back that a one-level-per-pass pyramid performs and this kernel does
            # not (SURVEY 8(d)'s algorithmic bytes); what the memory system really delivers is traffic / time:
            roof["achieved_traffic"] = round(roof["traffic"] / (med0 * frames_per_launch * 1e-6) / 1e9, 1)
            roof["frac_traffic"] = round(roof["achieved_traffic"] / HBM_PEAK_GBS, 4)
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / roof["bytes_per_launch"], 4)
            roof["note"] = ("achieved/frac: algorithmic bytes of levels 0+1 (SURVEY 8(d)) / median launch time; achieved_traffic/"
                            "frac_traffic: HBM bytes actually moved (PMC) / the same time.  Ablation builds (profiles/r03_pyramid_bounds.md): "
                            "the kernel's data flow alone runs at the memory system's ceiling for this mix (4.8-5.1 TB/s real traffic); since round 4 "
                            "(profiles/r04_pyramid_kernel.md) its arithmetic alone is 10 % shorter than that, together 9 % above the data-flow floor.")
        # ---- the temporal kernel K1 (unpack + display model + luminance + FIR -> pyramid level 0): its own roofline ----
        # algorithmic bytes per output frame: every source sample read once (2 streams x C channels x element size) and the
        # four temporal-channel planes written once (16 B per pixel); fl-1 history frames per launch on top
        es = pairs[0][0].element_size()
        Cc = pairs[0][0].shape[1]
        b_k1 = float(W * H) * (2.0 * Cc * es + 16.0)
        tk = t[:, 0]
        medk = float(np.median(tk))
        roof_k1 = {"bound": "hbm", "kernel": "temporal_vec_kernel<8,4,u8> (unpack + display-model LUT + RGB->Y + 8-tap temporal FIR of both "
                                            "streams -> pyramid level 0)" if (es, Cc, fps) == (1, 3, 30) else "temporal kernel (K1)",
                   "achieved": round(b_k1 / (medk * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(b_k1 / (medk * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                   "median_launch_ms": round(medk * frames_per_launch * 1e-3, 4), "min_launch_ms": round(float(np.min(tk)) * frames_per_launch * 1e-3, 4),
                   "max_launch_ms": round(float(np.max(tk)) * frames_per_launch * 1e-3, 4), "launches_timed": len(rows),
                   "bytes_per_launch": int(b_k1 * frames_per_launch), "frames_per_launch": frames_per_launch,
                   "bytes_per_pixel_frame": round(b_k1 / (W * H), 2),
                   "launch_ms_all": [round(float(x) * frames_per_launch * 1e-3, 4) for x in tk],
                   "note": "reads 6 B and writes 16 B per pixel and frame: nothing to save but the time.  How fast the 8 GB of level 0 can "
                           "be WRITTEN depends on where they lie physically (31-33 or 36-38 us per frame; a kernel that only replays the "
                           "store stream, and a plain streaming write, show the same two speeds on the same buffers; on the slow ones "
                           "the L2's write requests wait 6-20x longer for DRAM credits): profiles/r05_k1_mode.md.  The context keeps level 0 "
                           "in two ranges (even / odd frame slots) chosen at creation among half-size candidates: the pair that is written "
                           "fastest at once, i.e. two ranges of different kinds of physical memory where there are any (level0_alloc)."}
        if k1_live is not None and (W, H, int(frames_per_launch), es, Cc, fps) == (3840, 2160, 60, 1, 3, 30):
            roof_k1["traffic"] = int(k1_live["traffic_bytes"])
            roof_k1["traffic_source"] = ("measured in this run: the same two rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE) of the temporal "
                                         "kernel; includes the 7 history frames a launch reads before its first output (%.2f GB)" % (k1_live["history_bytes"] / 1e9))
            roof_k1["traffic_over_algorithmic"] = round(k1_live["traffic_bytes"] / roof_k1["bytes_per_launch"], 4)
        extra["roofline_k1"] = roof_k1
        extra["roofline_pyramid"] = roof
        extra["longest_kernel"] = "K1 temporal" if medk > med0 else "K2b pyramid levels 0+1"
        # graded pass = all band levels + finalize (B_alg of SURVEY section 8(d): 221.2 MB per 4K frame)
        tot = t[:, 1:].sum(axis=1)
        b_all = float(sum(alg))
        extra["graded_pass"] = {"levels_us_per_frame_median": [round(float(x), 2) for x in np.median(t[:, 1:n_bands + 1], axis=0)],
                                "finalize_us_per_frame": round(float(np.median(t[:, n_bands + 1])), 2),
                                "temporal_us_per_frame_median": round(float(np.median(t[:, 0])), 2),
                                "us_per_frame_all_levels": round(float(np.median(tot)), 2),
                                "us_per_frame_all_levels_min": round(float(np.min(tot)), 2),
                                "us_per_frame_all_levels_max": round(float(np.max(tot)), 2),
                                "algorithmic_bytes_per_frame": int(b_all),
                                "hbm_frac_all_levels": round(b_all / (float(np.median(tot)) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                "hbm_frac_all_levels_at_min": round(b_all / (float(np.min(tot)) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                "calls_timed": len(rows)}
        # SURVEY 8(d)'s conservative figure "reported alongside": the strict compulsory lower bound if nothing were materialised,
        # B_min = 4 * P * px_0 (level 0 read once; 132.7 MB per 4K frame), against the SAME time of the whole graded pass
        b_min = 4.0 * P * sizes[0][0] * sizes[0][1]
        extra["graded_pass"]["compulsory_bytes_per_frame_bmin"] = int(b_min)
        extra["graded_pass"]["hbm_frac_bmin"] = round(b_min / (float(np.median(tot)) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        roof["frac_bmin"] = extra["graded_pass"]["hbm_frac_bmin"]
        roof["frac_bmin_note"] = ("SURVEY 8(d) conservative figure: B_min = 4*P*px_0 = %.1f MB per frame (level 0 read once, nothing else "
                                  "credited) / the time of ALL pyramid levels + finalize (%.2f us per frame) / peak" % (b_min / 1e6, float(np.median(tot))))
        if foveated:
            # configs[3]: the whole graded pass of the foveated path (one-level kernels with per-pixel geometry) against SURVEY 8(d)'s bytes
            extra["roofline_fov"] = {"bound": "hbm",
                                     "note": "priced against the HBM peak as SURVEY 8(d) asks; the pass is in fact co-bound -- its arithmetic alone takes "
                                             "43.7 us per frame for all levels, its data flow alone ~44: profiles/r06_fov_floor.md",
                                     "kernel": "band_kernel<4, false, 1> x %d levels + finalize" % n_bands,
                                     "us_per_frame_levels": extra["graded_pass"]["levels_us_per_frame_median"],
                                     "us_per_frame_all_levels": extra["graded_pass"]["us_per_frame_all_levels"],
                                     "algorithmic_bytes_per_frame": int(b_all), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "achieved": round(b_all / (float(np.median(tot)) * 1e-6) / 1e9, 1),
                                     "frac": extra["graded_pass"]["hbm_frac_all_levels"], "frac_bmin": extra["graded_pass"]["hbm_frac_bmin"],
                                     "target_frac": 0.60, "target_us_per_frame": round(b_all / (0.6 * HBM_PEAK_GBS * 1e9) * 1e6, 2),
                                     "level0": {"us_per_frame": round(med0, 2), "algorithmic_bytes_per_frame": int(alg[0]),
                                                "frac": round(alg[0] / (med0 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                                     "calls_timed": len(rows), "frames_per_launch": frames_per_launch}
        # the headline kernel is the clamp-free variant the library may launch on SDR displays (proved per call: luminance_range /
        # clamps_never_bind); the variant WITH the clamps (HDR displays, sources that hand over their own luminance) beside it
        if fused01 and not foveated and world == 1 and K == 1 and not os.environ.get("FVVDP_BAND_INRANGE"):
            try:
                os.environ["FVVDP_BAND_INRANGE"] = "0"
                m2 = fv.fvvdp(display_name=args.display, device=dev)
                m2.timing = True
                rows2 = []
                for it in range(2 + max(6, args.timing_reps // 2)):
                    m2.predict(test, ref, dim_order="BCFHW", frames_per_second=fps)
                    nat.check(nat.lib().fvvdp_ctx_timing_read(m2._ctx.handle, ms, cnt, nk, 1))
                    if it >= 2:
                        rows2.append([ms[i] for i in range(n_bands + 2)])
                t2 = np.asarray(rows2, dtype=np.float64) / N * 1e3
                med2 = float(np.median(t2[:, 1]))
                roof["with_clamps"] = {"kernel": "band2_kernel<4, false> (the same launch with the four clamps kept: FVVDP_BAND_INRANGE=0)",
                                       "median_launch_ms": round(med2 * frames_per_launch * 1e-3, 4),
                                       "frac": round(b0 / (med2 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                       "us_per_frame_all_levels": round(float(np.median(t2[:, 1:].sum(axis=1))), 2),
                                       "hbm_frac_all_levels": round(b_all / (float(np.median(t2[:, 1:].sum(axis=1))) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                       "launches_timed": len(rows2)}
                del m2
            except Exception as e:
                sys.stderr.write("with_clamps leg failed (%s)\n" % e)
            finally:
                os.environ.pop("FVVDP_BAND_INRANGE", None)
        extra["batch_frames"] = batch
        # the whole step against its algorithmic bytes: source samples read once, the four temporal channels written once,
        # the pyramid by SURVEY 8(d) (K1 + graded pass), on the median step of the timed region (overlap included)
        b_step = (b_k1 + b_all) * N * K
        extra["roofline_step"] = {"bound": "hbm", "bytes_per_step": int(b_step), "ms_per_step": round(dt_med * 1e3, 3),
                                  "achieved": round(b_step / dt_med / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(b_step / dt_med / 1e9 / HBM_PEAK_GBS, 4),
                                  "kernels_back_to_back_ms": round(float(np.median(t.sum(axis=1))) * N * 1e-3, 3),
                                  "note": "bytes: (2*C*elem + 16) B per pixel and frame for the temporal kernel + SURVEY 8(d)'s pyramid "
                                          "bytes; kernels_back_to_back_ms: the sum of the isolated kernel medians of one pair"}

    # ---- the same call on pageable host arrays (PCIe-inclusive; reported beside `value`, never as `value`) ----
    if rank == 0 and world == 1 and not args.no_h2d:
        th, rh = test.cpu().numpy(), ref.cpu().numpy()
        m.predict(th, rh, dim_order="BCFHW", frames_per_second=fps, fixation_point=gz)
        torch.cuda.synchronize(dev)
        tb = []
        for _ in range(2):
            t1 = time.perf_counter()
            m.predict(th, rh, dim_order="BCFHW", frames_per_second=fps, fixation_point=gz)
            torch.cuda.synchronize(dev)
            tb.append(time.perf_counter() - t1)
        extra["value_h2d_inclusive"] = round(2.0 * W * H * N / min(tb) / 1e6, 1)
        extra["h2d_inclusive_note"] = "predict() on pageable numpy arrays (2 x %.2f GB uploaded per call), best of 2: %.1f ms" % (
            th.nbytes / 1e9, min(tb) * 1e3)
        del th, rh

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import fvvdp_oracle as orc
        tc, rc = test.cpu().numpy(), ref.cpu().numpy()
        fl = fl_guard(fps) + 1
        done = False
        if args.cpu_procs <= 0:
            args.cpu_procs = max(1, min(host_cpus_usable(), 64, host_mem_available_gb() // 3))
        if args.cpu_procs > 1:
            # the numpy oracle on several host cores: one output frame (with its own temporal window) per process,
            # all running concurrently (oracle/cpu_bench.py; plain subprocesses with a hard timeout)
            import tempfile
            from oracle import cpu_bench
            try:
                with tempfile.TemporaryDirectory() as d:
                    wall, per = cpu_bench.timed_frames(tc, rc, fps, args.display, fl, args.cpu_procs, d, gaze=(gaze[fl - 1] if foveated else None))
                cpu = {"value": round(2.0 * W * H * args.cpu_procs / wall / 1e6, 3), "unit": "Mpixels/s (test+ref)",
                       "cores": args.cpu_procs, "kind": "port",
                       "sample": "%d output frames (from frame %d on, each incl. its %d-frame temporal window) of the same %dx%d "
                                 "pair, one per process, numpy fp32 oracle, %d processes concurrently (%.1f - %.1f s per frame, "
                                 "%.1f s wall); host has %d cores (%s), of which this container may use %d (affinity / cgroup "
                                 "cpu.max).  Every process recomputes its frame's whole 8-frame temporal window (display model and "
                                 "luminance x8 per output frame), which a streaming CPU implementation would do once per frame: the "
                                 "figure understates what these cores could sustain -- a baseline, not a target" % (args.cpu_procs, fl - 1, fl, W, H, args.cpu_procs, min(per), max(per), wall,
                                               os.cpu_count(), host_cpu_model(), host_cpus_usable())}
                done = True
            except Exception as e:                      # never let the baseline leg break the benchmark line
                sys.stderr.write("cpu baseline: parallel run failed (%s), timing a single process instead\n" % e)
        if not done:
            nf = max(1, args.cpu_frames)
            frames = list(range(fl - 1, fl - 1 + nf))
            o = orc.Oracle(args.display, foveated=foveated)
            tcpu = time.perf_counter()
            o.predict(tc, rc, frames_per_second=fps, frames=frames, fixation_point=(gaze if foveated else None))
            tcpu = time.perf_counter() - tcpu
            cpu = {"value": round(2.0 * W * H * nf / tcpu / 1e6, 3), "unit": "Mpixels/s (test+ref)", "cores": 1,
                   "kind": "port", "sample": "%d output frames (frames %d..%d, incl. their %d-frame temporal window) of the same "
                   "%dx%d pair, numpy fp32 oracle, single thread; host has %d cores" % (nf, frames[0], frames[-1], fl, W, H, os.cpu_count())}

    # second half of the metric: |JOD - JOD of the reference| for rank 0's first pair, from the committed golden
    # (tests/golden/g3_synth_uhd_60f.npz = the reference's own torch-CPU run on this synthetic pair, tools/gen_golden.py g3)
    jod_delta = None
    gpath = os.path.join(ROOT, "tests", "golden", "g3_synth_uhd_60f.npz")
    if foveated:
        gpath = os.path.join(ROOT, "tests", "golden", "g4_foveated_uhd_120f.npz")
    if rank == 0 and os.path.exists(gpath) and not frames_mode and (
            (W, H, N, fps, args.display, foveated) in ((3840, 2160, 60, 30, "standard_4k", False), (3840, 2160, 120, 30, "standard_hdr_pq", True))):
        g = np.load(gpath)
        jod_delta = abs(float(jods[0]) - float(g["jod"]))
        if cpu is not None and "seconds" in g.files:
            # the real reference (PyTorch CPU, 8 threads) could only be timed in the build container (it cannot travel)
            cpu["reference_torch_cpu_seconds_build_container"] = round(float(g["seconds"]), 1)
            cpu["reference_torch_cpu_mpix_s_build_container"] = round(2.0 * W * H * N / float(g["seconds"]) / 1e6, 2)

    if rank == 0:
        out = {"metric": "Mpixels/s (test+ref) at 4Kx60f; JOD delta vs reference", "value": round(mpix, 1),
               "unit": "Mpixels/s (test+ref)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt_med * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%dx%d x%d-frame synthetic uint8 RGB video pair(s), %d per GPU and step, %s, %d fps, "
                                      "foveated=%s (%s)%s" % (W, H, n_clip, K, args.display, fps, "on, moving gaze" if foveated else "off",
                                                            baseline_label(W, H, N, fps, args.display, world, K, frames_mode, foveated),
                                                            "; SDR display: the pyramid pass runs its clamp-free variant band2_kernel<4, true> (the library "
                                                            "proves per call that the clamps cannot bind); the variant with clamps, which HDR displays take, "
                                                            "is in roofline.with_clamps" if (roof is not None and "with_clamps" in roof) else ""),
                          "pairs_per_gpu": K, "shard": args.shard,
                          "parallelism": ("frame-sharded x%d: one pair of %d frames, %d output frames per rank (+ fl-1 frames of temporal halo read "
                                          "by every rank from its own copy), one all-reduce of Q_per_ch on the device" % (world, n_clip, N))
                                         if frames_mode else
                                         ("pair-sharded x%d, %d pair(s) per rank queued without host sync, one all-reduce of the "
                                          "result rows on the device" % (world, K))},
               "step_path": step_path,
               "ms_per_pair": round(dt_med / K * 1e3, 3),
               "predict_call_ms": None if predict_ms is None else round(predict_ms, 3),
               "timing": {"statistic": "median of the %d timed steps (SURVEY 8(d)); value = pixels of a step / ms_per_step" % args.steps,
                          "ms_per_step_mean": round(dt / args.steps * 1e3, 3), "value_mean": round(mpix_mean, 1),
                          "ms_per_step_min": round(float(step_s[0]) * 1e3, 3), "ms_per_step_max": round(float(step_s[-1]) * 1e3, 3),
                          "ms_per_step_in_order": [round(float(x) * 1e3, 3) for x in tall[1:]],
                          "timed_region_s": round(dt, 4)},
               "jod": [round(j, 6) for j in jods[:8]], "jod_exact": [float(j).hex() for j in jods[:8]],
               "jod_delta_vs_reference": None if jod_delta is None else float("%.3g" % jod_delta), "roofline": roof, "cpu_baseline": cpu}
        out.update(extra)
        if comm is not None:
            out["communicator"] = comm
        if coll is not None:
            out["collective"] = coll
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()             # rank 0 ran the per-kernel timing, the PCIe leg and the CPU baseline alone: leave together
        dist.destroy_process_group()


def measure_traffic_live():
    """FETCH_SIZE and WRITE_SIZE of the dominant pyramid kernel, each in its own rocprofv3 --pmc pass (counters only with
    --kernel-trace, as MI355X_MICROARCH.md prescribes), on the same 4K x60 workload.  Returns the dict of tools/pmc_level0.py
    or None when rocprofv3 is not available / a pass fails (the caller then falls back to the committed profile)."""
    import glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None
    d = tempfile.mkdtemp(prefix="fvvdp_pmc_")
    try:
        dbs = []
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(d, ctr)
            subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", out, "-o", "p", "--", sys.executable,
                            os.path.join(ROOT, "tools", "gpu_bandonly.py")], check=True, timeout=120, cwd=d,
                           env=dict(os.environ, TMPDIR=d, STAGE="all", REPS="3"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            found = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if not found:
                return None
            dbs.append(found[0])
        res = os.path.join(d, "pmc.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_level0.py"), dbs[0], dbs[1], res], check=True,
                       timeout=60, stdout=subprocess.DEVNULL)
        with open(res) as f:
            live = json.load(f)
        # third pass, no counters: the plain kernel trace of the same target, 20 whole predict() calls -> the rocprof AVERAGE duration
        # of the dominant kernel (roofline.frac_rocprof_avg), cold launches included
        try:
            out = os.path.join(d, "trace")
            subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "p", "--", sys.executable,
                            os.path.join(ROOT, "tools", "gpu_bandonly.py")], check=True, timeout=180, cwd=d,
                           env=dict(os.environ, TMPDIR=d, STAGE="all", REPS="20"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            found = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if found:
                live["trace"] = dominant_kernel_trace(found[0], "band2_kernel" if "band2" in live.get("kernel", "") else "band_kernel")
        except Exception as e:
            sys.stderr.write("kernel-trace pass failed (%s); roofline.frac_rocprof_avg omitted\n" % e)
        return live
    except Exception as e:
        sys.stderr.write("live traffic measurement failed (%s); using the committed profile\n" % e)
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def dominant_kernel_trace(db, substr):
    """avg / median / steady median (first 3 dropped) of the LARGE dispatches of the kernels whose name contains `substr` in a
    rocprofv3 rocpd database (the two-level kernel is launched twice per batch at 4K: levels 0+1 and 2+3; the large ones are 0+1)"""
    import sqlite3
    con = sqlite3.connect(db)
    rows = con.execute("select name, duration from kernels where name like ? order by start", ("%" + substr + "%",)).fetchall()
    con.close()
    if not rows:
        return None
    # the timed steps launch ONE instantiation; fvvdp_ctx_create times its level-0 layout once with the variant that keeps the clamps
    # (3 launches of <P, false> on a synthetic clip, first touch included): where both appear, the clamp-free <P, true> is the steps'
    names = sorted(set(r[0] for r in rows))
    pick = [n for n in names if "true>" in n] or names
    d = [r[1] for r in rows if r[0] in pick]
    big = [x for x in d if x > 0.5 * max(d)]
    steady = big[3:] if len(big) > 3 else big
    return {"kernel": pick[0], "n": len(big), "avg_us": float(np.mean(big)) / 1e3, "median_us": float(np.median(big)) / 1e3,
            "steady_median_us": float(np.median(steady)) / 1e3, "min_us": float(np.min(big)) / 1e3, "max_us": float(np.max(big)) / 1e3}


def host_cpus_usable():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max / cfs_quota_us) -- the GPU
    boxes show all 256 cores of the host but grant the container 16 of them (64 oracle processes ran 15x slower each)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def host_mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) // (1 << 20)
    except OSError:
        pass
    return 16


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def baseline_label(W, H, N, fps, display, world, K, frames_mode, foveated=False):
    """which BASELINE.json configs[] entry the arguments describe (the label is derived, never assumed)"""
    if foveated:
        return ("BASELINE.json configs[3]" if (W, H, N, fps, display, frames_mode) == (3840, 2160, 120, 30, "standard_hdr_pq", False)
                else "not a BASELINE.json configuration")
    if frames_mode:
        return "frame sharding of one long pair: north_star's frame-parallel decomposition, not a configs[] entry"
    if (W, H, N, fps, display) == (3840, 2160, 60, 30, "standard_4k"):
        return "BASELINE.json configs[2]" + ("; configs[4]: 64 pairs over 8 GPUs" if (world, K) == (8, 8) else
                                            ("; %d of configs[4]'s 64 pairs per step" % (world * K) if world * K > 1 else ""))
    if (W, H, N, fps, display) == (1920, 1080, 60, 30, "standard_fhd"):
        return "BASELINE.json configs[1]"
    return "not a BASELINE.json configuration"


def fl_guard(fps):
    return int(np.ceil(250.0 / (1000.0 / fps))) - 1


if __name__ == "__main__":
    main()
