"""Multi-GPU sharding of the path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).

The path shards into independent units with no data-path exchange:
  * pair sharding  - independent test/reference pairs, one contiguous block of pairs per rank;
  * frame sharding - output frames [f0, f1) of one video per rank; frame f only needs source frames f-fl+1..f,
                     which every rank reads from its own copy of the source (temporal halo, no exchange).
The only collective is ONE all-reduce(sum) of the per-frame pooled values Q_per_ch: every rank writes its own
(pair, frame) slots of a zero-initialised buffer, so the sum of disjoint supports is an all-gather and directly
yields the operand of the temporal pool / JOD regression.  The payload is KBs (4Kx60: 3.4 KB per pair), i.e.
latency-bound; xGMI link bandwidth is irrelevant here.
"""
import torch


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of `n_items` for `rank` (the first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _all_reduce_sum(buf, group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def gather_pair_results(q_local, rank, world, group=None):
    """q_local: this rank's Q_per_ch [bands, 2, N] (or a stack [k, ...] of k result rows per rank -- Q_per_ch tensors or the flat
    `stats['result_buffer']` rows of `predict(..., sync=False)` --, equal k on all ranks) -> [world * k, ...] on every rank through
    one all-reduce."""
    if q_local.dim() == 3:
        q_local = q_local.unsqueeze(0)
    if world == 1:
        return q_local                         # one rank: its rows are the result (no buffer, no collective)
    k = q_local.shape[0]
    buf = torch.zeros((world * k,) + tuple(q_local.shape[1:]), dtype=q_local.dtype, device=q_local.device)
    buf[rank * k:(rank + 1) * k] = q_local
    return _all_reduce_sum(buf, group)


def frame_sharded_q(compute_range, n_bands, n_frames, rank, world, device, group=None, dtype=torch.float32):
    """Frame sharding of one video.  `compute_range(f0, f1)` returns Q_per_ch[:, :, f0:f1] (tensor [bands,2,f1-f0])
    for this rank's frames; returns the complete Q_per_ch [bands, 2, n_frames] on every rank."""
    f0, f1 = shard_range(n_frames, rank, world)
    buf = torch.zeros((n_bands, 2, n_frames), dtype=dtype, device=device)
    if f1 > f0:
        buf[:, :, f0:f1] = compute_range(f0, f1).to(device=device, dtype=dtype)
    return _all_reduce_sum(buf, group)


def predict_frame_sharded(metric, vid_source, rank, world, fixation_point=None, group=None):
    """Frame-sharded `fvvdp.predict_video_source`: every rank returns the same (Q_JOD, stats).  This rank's frames
    are queued without a host synchronisation; their per-frame results go from the kernels' output buffer straight into
    this rank's slots of the all-reduce buffer on the device."""
    H, W, N = vid_source.get_video_size()
    holder = {}

    def compute(f0, f1):
        _, stats = metric.predict_video_source(vid_source, fixation_point=fixation_point, frame_range=(f0, f1), pool=False,
                                               sync=False)
        holder["stats"] = stats
        return stats["Q_per_ch"]

    from .fvvdp import band_frequencies
    n_bands, rho_band = band_frequencies(W, H, metric.pix_per_deg)
    Q = frame_sharded_q(compute, n_bands, N, rank, world, metric.device, group)
    stats = holder.get("stats")
    if stats is None:                       # more ranks than frames: this rank computed nothing
        stats = {"rho_band": rho_band, "frames_per_second": vid_source.get_frames_per_second(), "width": W, "height": H,
                 "N_frames": N}
    # any rank saw an out-of-range pixel -> every rank warns, like the unsharded call.  EVERY rank takes part in the
    # reduce, also one whose shard is empty (more ranks than frames: it computed nothing and has no flag of its own);
    # a collective that only some ranks call hangs or pairs with a later one.
    flag = stats.pop("range_flag", None)
    flag = torch.zeros(1, dtype=torch.int32, device=metric.device) if flag is None else flag.clone()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    jod = metric.do_pooling_and_jods(Q, None)
    stats["Q_per_ch"] = Q.cpu().numpy()
    if int(flag.cpu()[0]) != 0:
        import logging
        logging.warning("Pixel outside the valid range 0-1")
    return jod, stats
