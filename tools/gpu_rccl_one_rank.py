#!/usr/bin/env python3
"""RCCL on ONE GPU through the product's step (target of `rocprofv3 --kernel-trace`, profiles/r06_rccl_single_rank.md).

A world-size-1 `nccl` process group (in-process store, device_id = this GPU).  The result rows of `predict(..., sync=False)` -- written by
the library's kernels on the caller's stream -- go through `gather_pair_results(force_collective=True)`, i.e. ncclAllReduce(sum) in place on
a zero-initialised buffer, exactly what every rank does at N > 1.  On one rank RCCL's in-place sum needs no device work (it returns after
the stream hand-over: no kernel shows up in a trace); the same buffer through ReduceOp.AVG takes RCCL's one-rank reduce kernel
(PreMulSum with scalar 1/1), which DOES show up -- RCCL device code running on this GPU, on data the library produced.  Both results must be
bit-equal to the rows.  Prints the host cost per call of each and the RCCL version."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fovvideovdp_amd as fv
from fovvideovdp_amd.sharding import gather_pair_results
from fovvideovdp_amd.synth import synth_video_pair

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", store=dist.HashStore(), rank=0, world_size=1, device_id=dev)
H, W, N = [int(v) for v in os.environ.get("HWN", "2160,3840,60").split(",")]
K = int(os.environ.get("PAIRS", "2"))
pairs = [synth_video_pair(N, H, W, device=dev, pair=k) for k in range(K)]
m = fv.fvvdp(display_name="standard_4k", device=dev)
for it in range(3):
    outs = m.predict_batch(pairs, frames_per_second=30)
    rows = torch.stack([st["result_buffer"] for (_, st) in outs])
    got = gather_pair_results(rows, 0, 1, force_collective=True)
    avg = rows.clone()
    dist.all_reduce(avg, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    assert got.data_ptr() != rows.data_ptr()
    same_sum = bool(torch.equal(got.view(torch.int32), rows.view(torch.int32)))
    same_avg = bool(torch.equal(avg.view(torch.int32), rows.view(torch.int32)))
    print("step %d: %d rows x %d floats, JOD %s | all_reduce(sum) bit-equal to the rows: %s | all_reduce(avg) bit-equal: %s" % (
        it, rows.shape[0], rows.shape[1], [round(float(v), 6) for v in got[:, -1].cpu()], same_sum, same_avg))
    assert same_sum and same_avg
for name, op in (("sum", dist.ReduceOp.SUM), ("avg", dist.ReduceOp.AVG)):
    buf = rows.clone()
    for _ in range(10):
        dist.all_reduce(buf, op=op)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        dist.all_reduce(buf, op=op)
    torch.cuda.synchronize()
    print("all_reduce(%s), %d B, in place, back to back: %.1f us per call (host clock around %d calls + one synchronise)" % (
        name, buf.numel() * 4, (time.perf_counter() - t0) / n * 1e6, n))
print("backend", dist.get_backend(), "world", dist.get_world_size(), "rccl", ".".join(str(v) for v in torch.cuda.nccl.version()),
      "device", torch.cuda.get_device_name(dev))
dist.barrier()
dist.destroy_process_group()
