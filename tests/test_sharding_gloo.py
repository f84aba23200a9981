"""Multi-process tests of the sharding layer on CPU (gloo, world_size 2): the one collective of the path is an
all-reduce(sum) of disjoint Q_per_ch slots.  The per-rank compute is the oracle here (tests only); on the GPU the
same code path runs with the HIP metric and backend nccl (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fovvideovdp_amd.sharding import shard_range, gather_pair_results, frame_sharded_q


def test_shard_range_partitions():
    for n in (1, 7, 60, 64, 121):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import fvvdp_oracle as orc
        from fovvideovdp_amd.synth import synth_video_pair
        import fovvideovdp_amd as fv
        N, H, W, fps = 9, 40, 64, 30
        test, ref = synth_video_pair(N, H, W)
        t, r = test.numpy(), ref.numpy()
        o = orc.Oracle("standard_fhd")

        def compute(f0, f1):
            _, st = o.predict(t, r, frames_per_second=fps, frames=range(f0, f1))
            return torch.from_numpy(st["Q_per_ch"])

        n_bands = orc.band_frequencies(W, H, o.ppd)[0]
        Q = frame_sharded_q(compute, n_bands, N, rank, world, torch.device("cpu"))
        m = fv.fvvdp(display_name="standard_fhd", device=torch.device("cpu"))
        jod = float(m.do_pooling_and_jods(Q, None))
        # pair sharding: each rank owns one pair
        tp, rp = synth_video_pair(4, H, W, pair=rank)
        _, stp = o.predict(tp.numpy(), rp.numpy(), frames_per_second=fps)
        allq = gather_pair_results(torch.from_numpy(stp["Q_per_ch"]), rank, world)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), Q=Q.numpy(), jod=jod, allq=allq.numpy())
    finally:
        dist.destroy_process_group()


def test_frame_and_pair_sharding_world2(tmp_path):
    from oracle import fvvdp_oracle as orc
    from fovvideovdp_amd.synth import synth_video_pair
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    N, H, W, fps = 9, 40, 64, 30
    test, ref = synth_video_pair(N, H, W)
    jod, st = orc.Oracle("standard_fhd").predict(test.numpy(), ref.numpy(), frames_per_second=fps)
    for k in range(world):
        assert np.array_equal(res[k]["Q"], st["Q_per_ch"])          # sharded == unsharded, on every rank
        assert abs(float(res[k]["jod"]) - float(jod)) < 2e-6
    assert np.array_equal(res[0]["allq"], res[1]["allq"])
    for k in range(world):
        tp, rp = synth_video_pair(4, H, W, pair=k)
        _, stp = orc.Oracle("standard_fhd").predict(tp.numpy(), rp.numpy(), frames_per_second=fps)
        assert np.array_equal(res[0]["allq"][k], stp["Q_per_ch"])
