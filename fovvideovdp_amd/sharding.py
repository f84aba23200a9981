"""Multi-GPU sharding of the path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on
MI355X; "gloo" in the CPU tests).

The path shards into independent units with no data-path exchange:
  * pair sharding  - independent test/reference pairs, one contiguous block of pairs per rank;
  * frame sharding - output frames [f0, f1) of one video per rank; frame f only needs source frames f-fl+1..f,
                     which every rank reads from its own copy of the source (temporal halo, no exchange).
The only collective is ONE all-reduce(sum) of the per-frame pooled values Q_per_ch (with the out-of-range flag riding in
the same buffer): every rank writes its own (pair, frame) slots of a zero-initialised buffer, so the sum of disjoint
supports is an all-gather and directly yields the operand of the temporal pool / JOD regression.  The payload is KBs (4Kx60: 3.4 KB per pair), i.e.
latency-bound; xGMI link bandwidth is irrelevant here.
"""
import torch


def shard_range(n_items, rank, world):
    """Contiguous, balanced [lo, hi) of `n_items` for `rank` (the first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def collective_active(group=None, force=False):
    """True when the all-reduce of the path is really issued: a process group exists and it has more than one rank -- or the
    caller insists (`force`: bench.py at N = 1 runs the SAME step as N > 1, so RCCL is initialised, the buffer goes through
    `ncclAllReduce` on one rank, and the N = 1 point of a scaling curve carries the collective's fixed cost too)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return force or dist.get_world_size(group) > 1


def _all_reduce_sum(buf, group=None, force=False):
    if collective_active(group, force):
        import torch.distributed as dist
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def gather_pair_results(q_local, rank, world, group=None, force_collective=False):
    """q_local: this rank's Q_per_ch [bands, 2, N] (or a stack [k, ...] of k result rows per rank -- Q_per_ch tensors or the flat
    `stats['result_buffer']` rows of `predict(..., sync=False)` --, equal k on all ranks) -> [world * k, ...] on every rank through
    one all-reduce.  One rank without `force_collective`: the rows come back as they are (no buffer, no collective)."""
    if q_local.dim() == 3:
        q_local = q_local.unsqueeze(0)
    if world == 1 and not collective_active(group, force_collective):
        return q_local
    k = q_local.shape[0]
    buf = torch.zeros((world * k,) + tuple(q_local.shape[1:]), dtype=q_local.dtype, device=q_local.device)
    buf[rank * k:(rank + 1) * k] = q_local
    return _all_reduce_sum(buf, group, force_collective)


def frame_sharded_row(compute_range, n_bands, n_frames, rank, world, device, group=None, dtype=torch.float32,
                      force_collective=False):
    """Frame sharding of one video through ONE collective.  `compute_range(f0, f1)` returns this rank's
    (Q_per_ch[:, :, f0:f1] as a tensor [bands, 2, f1-f0], out-of-range flag: int tensor [1] or None).  The all-reduced row is
    [Q_per_ch (bands x 2 x n_frames) | number of ranks that saw an out-of-range pixel]: returns (Q_per_ch [bands, 2, n_frames],
    the flat row) on every rank; row[-1] != 0 <=> some rank's flag was set."""
    f0, f1 = shard_range(n_frames, rank, world)
    nq = n_bands * 2 * n_frames
    row = torch.zeros(nq + 1, dtype=dtype, device=device)
    Q = row[:nq].view(n_bands, 2, n_frames)
    if f1 > f0:
        q, flag = compute_range(f0, f1)
        Q[:, :, f0:f1] = q.to(device=device, dtype=dtype)
        if flag is not None:
            row[nq:] = (flag.to(device).reshape(-1)[:1] != 0).to(dtype)
    _all_reduce_sum(row, group, force_collective)
    return Q, row


def frame_sharded_q(compute_range, n_bands, n_frames, rank, world, device, group=None, dtype=torch.float32):
    """As frame_sharded_row for a `compute_range(f0, f1)` that returns Q_per_ch[:, :, f0:f1] alone; returns the complete
    Q_per_ch [bands, 2, n_frames] on every rank."""
    return frame_sharded_row(lambda a, b: (compute_range(a, b), None), n_bands, n_frames, rank, world, device, group, dtype)[0]


def predict_frame_sharded(metric, vid_source, rank, world, fixation_point=None, group=None, force_collective=False):
    """Frame-sharded `fvvdp.predict_video_source`: every rank returns the same (Q_JOD, stats).  This rank's frames
    are queued without a host synchronisation; their per-frame results and the out-of-range flag go from the kernels' output
    buffer straight into this rank's slots of the ONE all-reduce buffer on the device (a rank whose shard is empty -- more
    ranks than frames -- computed nothing and still takes part: a collective that only some ranks call hangs or pairs with a
    later one); one device -> host copy of the reduced row."""
    H, W, N = vid_source.get_video_size()
    holder = {}

    def compute(f0, f1):
        _, stats = metric.predict_video_source(vid_source, fixation_point=fixation_point, frame_range=(f0, f1), pool=False,
                                               sync=False)
        holder["stats"] = stats
        return stats["Q_per_ch"], stats.get("range_flag")

    from .fvvdp import band_frequencies
    n_bands, rho_band = band_frequencies(W, H, metric.pix_per_deg)
    Q, row = frame_sharded_row(compute, n_bands, N, rank, world, metric.device, group, force_collective=force_collective)
    stats = holder.get("stats")
    if stats is None:                       # more ranks than frames: this rank computed nothing
        stats = {"rho_band": rho_band, "frames_per_second": vid_source.get_frames_per_second(), "width": W, "height": H,
                 "N_frames": N}
    stats.pop("range_flag", None)
    stats.pop("result_buffer", None)
    jod = metric.do_pooling_and_jods(Q, None)
    host = row.cpu()                        # Q_per_ch and the flag count in one copy
    stats["Q_per_ch"] = host[:-1].view(n_bands, 2, N).numpy()
    if float(host[-1]) != 0:                # any rank saw an out-of-range pixel -> every rank warns, like the unsharded call
        import logging
        logging.warning("Pixel outside the valid range 0-1")
    return jod, stats
