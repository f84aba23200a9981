"""GPU: the multi-rank path with the HIP metric.  Two `gloo` ranks share the one GPU of the test box (the same code
runs with backend nccl = RCCL, one rank per GPU, under bench.py --gpus N): frame sharding through
`predict_frame_sharded`, pair sharding with several pairs per rank queued without host synchronisation
(`predict(..., sync=False)` -> device Q_per_ch -> `gather_pair_results`), both bit-equal to the unsharded calls."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, H, W, FPS = 11, 72, 130, 30
PAIRS_PER_RANK = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fovvideovdp_amd as fv
        from fovvideovdp_amd.sharding import gather_pair_results, predict_frame_sharded
        from fovvideovdp_amd.synth import synth_video_pair
        dev = torch.device("cuda", 0)
        m = fv.fvvdp(display_name="standard_fhd", device=dev)
        test, ref = synth_video_pair(N, H, W, device=dev)
        vs = fv.fvvdp_video_source_array(test, ref, FPS, display_photometry=m.display_photometry)
        jod, st = predict_frame_sharded(m, vs, rank, world)
        # pair sharding, PAIRS_PER_RANK pairs per rank, no host synchronisation between the pairs
        qs = []
        for k in range(PAIRS_PER_RANK):
            tp, rp = synth_video_pair(N, H, W, device=dev, pair=rank * PAIRS_PER_RANK + k)
            _, s = m.predict(tp, rp, frames_per_second=FPS, sync=False)
            assert isinstance(s["Q_per_ch"], torch.Tensor) and s["Q_per_ch"].is_cuda
            qs.append(s["Q_per_ch"])
        allq = gather_pair_results(torch.stack(qs), rank, world)
        jods = m.do_pooling_and_jods(allq, None)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), jod=float(jod), Q=st["Q_per_ch"], allq=allq.cpu().numpy(),
                 jods=jods.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu(tmp_path):
    import torch.multiprocessing as mp
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    m = fv.fvvdp(display_name="standard_fhd")
    test, ref = synth_video_pair(N, H, W, device="cuda")
    q, st = m.predict(test, ref, frames_per_second=FPS)
    for k in range(world):
        assert np.array_equal(res[k]["Q"], st["Q_per_ch"])            # frame-sharded == unsharded, bit for bit
        assert abs(float(res[k]["jod"]) - float(q)) < 2e-6
    assert np.array_equal(res[0]["allq"], res[1]["allq"]) and np.array_equal(res[0]["jods"], res[1]["jods"])
    assert res[0]["allq"].shape[0] == world * PAIRS_PER_RANK
    for p in range(world * PAIRS_PER_RANK):
        tp, rp = synth_video_pair(N, H, W, device="cuda", pair=p)
        qp, sp = m.predict(tp, rp, frames_per_second=FPS)
        assert np.array_equal(res[0]["allq"][p], sp["Q_per_ch"])
        assert abs(float(res[0]["jods"][p]) - float(qp)) < 2e-6


def test_sync_false_and_finish(caplog):
    import logging
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    m = fv.fvvdp(display_name="standard_fhd")
    test, ref = synth_video_pair(6, 64, 96, device="cuda")
    q0, st0 = m.predict(test, ref, frames_per_second=FPS)
    q1, st1 = m.predict(test, ref, frames_per_second=FPS, sync=False)
    assert st1["Q_per_ch"].is_cuda and "range_flag" in st1
    fv.fvvdp.finish(st1)
    assert isinstance(st1["Q_per_ch"], np.ndarray) and np.array_equal(st1["Q_per_ch"], st0["Q_per_ch"])
    assert float(q1) == float(q0)
    # the deferred out-of-range warning
    t = test.to(torch.float32) / 255 * 1.5
    with caplog.at_level(logging.WARNING):
        _, st2 = m.predict(t, ref.to(torch.float32) / 255, frames_per_second=FPS, sync=False)
        assert not any("outside the valid range" in r.message for r in caplog.records)
        fv.fvvdp.finish(st2)
    assert any("outside the valid range" in r.message for r in caplog.records)


def test_use_checkpoints_flag_is_accepted_and_gradients_are_refused():
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    m = fv.fvvdp(display_name="standard_fhd", use_checkpoints=True)
    test, ref = synth_video_pair(3, 40, 64, device="cuda")
    q, _ = m.predict(test, ref, frames_per_second=FPS)
    q2, _ = fv.fvvdp(display_name="standard_fhd").predict(test, ref, frames_per_second=FPS)
    assert float(q) == float(q2)
    t = (test.to(torch.float32) / 255).requires_grad_(True)
    with pytest.raises(RuntimeError, match="[Gg]radients"):
        m.predict(t, ref.to(torch.float32) / 255, frames_per_second=FPS)
    with torch.no_grad():
        m.predict(t, ref.to(torch.float32) / 255, frames_per_second=FPS)


def _run_bench(backend, world, extra=(), dims=(960, 540, 20, 2), torchrun=None):
    """bench.py launched the way the driver launches it (torch.distributed.run, one process per rank; a plain process for one rank);
    returns the JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    W, H, N, K = dims
    args = [os.path.join(root, "bench.py"), "--gpus", str(world), "--backend", backend,
            "--pairs-per-gpu", str(K), "--width", str(W), "--height", str(H), "--frames", str(N), "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline", "--no-h2d", "--no-measure-traffic"] + list(extra)
    if world > 1 or torchrun:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    else:
        cmd = [sys.executable] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=840)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                   # ONE JSON line, from rank 0
    return json.loads(lines[0]), (W, H, N, K)


def _check_bench_line(out, dims, backend, world=2):
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    W, H, N, K = dims
    assert out["n_gpus"] == world and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["pairs_per_gpu"] == K and out["value"] > 0 and out["higher_is_better"] is True
    assert out["config"]["shard"] == "pairs" and out["step_path"].startswith("predict_batch")     # ONE step path for every N
    assert "not a BASELINE.json configuration" in out["config"]["workload"]                       # the label follows the arguments
    assert out["roofline"] is not None and out["roofline"]["bound"] == "hbm"
    assert out["roofline_k1"]["bound"] == "hbm" and out["roofline_k1"]["frac"] > 0
    assert out["level0_alloc"]["per_frame_calls"] == {"host_syncs": 0, "allocations": 0, "frees": 0}
    # value is the median step; the mean is carried beside it, and the two are consistent with their times
    tm = out["timing"]
    px = 2.0 * W * H * N * K * world
    # (the line carries the times rounded to a microsecond: half a unit of that on a step of a few hundred microseconds counts)
    assert abs(out["value"] - px / (out["ms_per_step"] * 1e-3) / 1e6) <= (1e-3 + 0.0006 / out["ms_per_step"]) * out["value"]
    assert abs(tm["value_mean"] - px / (tm["ms_per_step_mean"] * 1e-3) / 1e6) <= (1e-3 + 0.0006 / tm["ms_per_step_mean"]) * tm["value_mean"]
    assert tm["ms_per_step_min"] <= out["ms_per_step"] <= tm["ms_per_step_max"]
    assert len(tm["ms_per_step_in_order"]) == out["steps"] and max(tm["ms_per_step_in_order"]) == tm["ms_per_step_max"]
    if world > 1:
        # what the communicator reports: backend, size, one entry per rank
        cm = out["communicator"]
        assert cm["backend"] == backend and cm["world_size"] == world and [r["rank"] for r in cm["ranks"]] == list(range(world))
        assert len(set(r["pid"] for r in cm["ranks"])) == world
        if backend == "nccl":
            assert cm["distinct_devices"] == world and cm["rccl_version"]
        else:
            assert cm["distinct_devices"] == min(world, torch.cuda.device_count())      # dry run: the ranks share the GPUs there are
        assert out["predict_call_ms"] is None
    else:
        assert (out["predict_call_ms"] > 0) if K == 1 else (out["predict_call_ms"] is None)     # the reference-style call beside a one-pair step
        cm = out.get("communicator")
        if cm is not None and cm.get("backend"):            # --collective auto / force: a world-size-1 group of the backend
            assert cm["backend"] == backend and cm["world_size"] == 1 and "forced" in cm["collective_on_one_rank"]
            assert "world-size-1 %s group" % backend in out["step_path"] and out["collective"]["us_per_call_back_to_back"] > 0
        else:                                               # --collective off, or no communicator on this box (reason in the line)
            assert "skipped" in out["step_path"] and "collective" not in out
    assert len(out["jod"]) == min(8, world * K)
    m = fv.fvvdp(display_name="standard_4k")
    for pidx in range(len(out["jod"])):
        t, r = synth_video_pair(N, H, W, device="cuda", pair=pidx)
        q, _ = m.predict(t, r, frames_per_second=30)
        assert abs(float(q) - out["jod"][pidx]) < 2e-6, (pidx, float(q), out["jod"][pidx])


@pytest.mark.timeout(600)
def test_bench_multi_rank_step_under_gloo():
    """bench.py's step (pairs queued with predict_batch -> all-reduce of the result rows -> one device-to-host copy), launched the
    way the driver launches it (torch.distributed.run, one process per rank), here with 2 gloo ranks sharing the GPU.  The
    JSON line must carry n_gpus 2, the communicator's own facts, median and mean timing, and the 4 JODs of the 4 pairs,
    equal to single-rank calls on the same pairs."""
    out, dims = _run_bench("gloo", 2)
    _check_bench_line(out, dims, "gloo")


@pytest.mark.timeout(600)
def test_bench_one_rank_takes_the_same_step_path():
    """N = 1 runs the SAME step path as N > 1 (predict_batch + the gather, a no-op on one rank, + one copy) -- the scaling curve
    compares like with like -- and reports the reference-style synchronous predict() once beside it (`predict_call_ms`)."""
    out, dims = _run_bench("nccl", 1, dims=(960, 540, 20, 1))
    _check_bench_line(out, dims, "nccl", world=1)


@pytest.mark.timeout(900)
def test_rccl_executes_on_one_rank_through_the_step_path():
    """RCCL on hardware through the product's own step.  bench.py launched the way the driver launches a multi-rank
    run (torch.distributed.run, here --nproc-per-node 1) with `--collective force`: init_process_group("nccl", world_size=1, device_id),
    every step's result rows (written by the library's kernels on the caller's stream) go through the zero-buffer all-reduce
    WITHOUT the one-rank shortcut, and the line carries what the communicator reports.  The JODs are bit-equal to the same run
    with `--collective off` (the shortcut), `--collective auto` (the default of a plain `python bench.py`) takes the forced path
    on a box where RCCL initialises, and frame sharding issues its one collective the same way."""
    import subprocess
    import sys
    probe = subprocess.run([sys.executable, "-c",
                            "import torch, torch.distributed as d\n"
                            "dev = torch.device('cuda', 0); torch.cuda.set_device(dev)\n"
                            "d.init_process_group('nccl', store=d.HashStore(), rank=0, world_size=1, device_id=dev)\n"
                            "t = torch.ones(4, device=dev); d.all_reduce(t); torch.cuda.synchronize(); assert float(t.sum()) == 4.0\n"
                            "d.destroy_process_group()\n"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    if probe.returncode != 0:          # an environment without a usable librccl: nothing of THIS repository to test
        pytest.skip("RCCL cannot create a one-rank communicator on this box: " + probe.stderr[-300:])
    dims = (960, 540, 20, 2)
    forced, _ = _run_bench("nccl", 1, extra=("--collective", "force"), dims=dims, torchrun=True)
    cm = forced["communicator"]
    assert cm["backend"] == "nccl" and cm["world_size"] == 1 and cm["rccl_version"] and cm["distinct_devices"] == 1
    assert cm["ranks"][0]["device_uuid"] and "forced" in cm["collective_on_one_rank"]
    assert forced["collective"]["backend"] == "nccl" and 0 < forced["collective"]["us_per_call_back_to_back"] < 2000
    _check_bench_line(forced, dims, "nccl", world=1)
    off, _ = _run_bench("nccl", 1, extra=("--collective", "off"), dims=dims)
    assert "communicator" not in off and "collective" not in off
    assert forced["jod_exact"] == off["jod_exact"] and len(off["jod_exact"]) == 2          # bit-equal, both pairs
    auto, _ = _run_bench("nccl", 1, dims=dims)                                          # what the driver's N = 1 run does
    assert auto["communicator"]["backend"] == "nccl" and auto["jod_exact"] == off["jod_exact"]
    fr_forced, _ = _run_bench("nccl", 1, extra=("--shard", "frames", "--collective", "force"), dims=(960, 540, 12, 1), torchrun=True)
    fr_off, _ = _run_bench("nccl", 1, extra=("--shard", "frames", "--collective", "off"), dims=(960, 540, 12, 1))
    assert fr_forced["communicator"]["backend"] == "nccl" and fr_forced["jod_exact"] == fr_off["jod_exact"]


@pytest.mark.timeout(600)
def test_bench_one_rank_without_a_communicator_falls_back_and_says_so():
    """`--collective auto` (the default) on a box where the backend cannot create a communicator (here: a backend name that does not
    exist): the step takes the one-rank shortcut, the line says why (`communicator.error`) and the benchmark still completes."""
    out, dims = _run_bench("no_such_backend", 1, dims=(960, 540, 20, 1))
    cm = out["communicator"]
    assert cm["backend"] is None and cm["error"] and "shortcut" in cm["collective_on_one_rank"]
    _check_bench_line(out, dims, "no_such_backend", world=1)


@pytest.mark.timeout(900)
def test_bench_eight_ranks_dry_run_under_gloo():
    """The 8-rank launch of the round-end scaling run, dry: 8 gloo ranks sharing this box's GPU(s) at reduced size, one pair each."""
    out, dims = _run_bench("gloo", 8, dims=(480, 270, 12, 1))
    _check_bench_line(out, dims, "gloo", world=8)


@pytest.mark.timeout(600)
def test_bench_frame_sharded_under_gloo():
    """`--shard frames`: ONE pair of --frames x N frames, every rank evaluates its own output frames (+ fl-1 frames of halo) and one
    all-reduce of Q_per_ch completes the clip on every rank.  JOD equal to the unsharded call on the same clip."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    W, H, N = 960, 540, 12
    out, _ = _run_bench("gloo", 2, extra=("--shard", "frames"), dims=(W, H, N, 1))
    assert out["config"]["shard"] == "frames" and out["step_path"].startswith("predict_frame_sharded") and out["scaling"] == "weak"
    assert "one pair of %d frames" % (2 * N) in out["config"]["parallelism"]
    px = 2.0 * W * H * N * 2
    assert abs(out["value"] - px / (out["ms_per_step"] * 1e-3) / 1e6) <= 2e-3 * out["value"]
    t, r = synth_video_pair(2 * N, H, W, device="cuda")
    q, _ = fv.fvvdp(display_name="standard_4k").predict(t, r, frames_per_second=30)
    assert len(out["jod"]) == 1 and abs(float(q) - out["jod"][0]) < 2e-6


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL run needs two GPUs (the driver's multi-GPU box)")
def test_bench_multi_rank_step_under_rccl():
    """The same launch with backend nccl (= RCCL over xGMI), one GPU per rank: init with device_id, the all-reduce of the
    device buffer, the barrier before destroy_process_group.  Skipped on single-GPU boxes."""
    out, dims = _run_bench("nccl", 2)
    _check_bench_line(out, dims, "nccl")


def _worker8(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fovvideovdp_amd as fv
        from fovvideovdp_amd.sharding import predict_frame_sharded, shard_range
        from fovvideovdp_amd.synth import synth_video_pair
        dev = torch.device("cuda", rank % torch.cuda.device_count())
        m = fv.fvvdp(display_name="standard_fhd", device=dev)
        res = {}
        for n in (5, 19):                                        # 5 frames over 8 ranks: three ranks have an empty shard
            test, ref = synth_video_pair(n, 54, 96, device=dev)
            vs = fv.fvvdp_video_source_array(test, ref, FPS, display_photometry=m.display_photometry)
            jod, st = predict_frame_sharded(m, vs, rank, world)
            res["jod%d" % n], res["Q%d" % n] = float(jod), st["Q_per_ch"]
            res["share%d" % n] = np.asarray(shard_range(n, rank, world))
        np.savez(os.path.join(out_dir, f"r8_{rank}.npz"), **res)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_ranks_frame_sharding_with_empty_shards(tmp_path):
    """World 8 (the node the path is meant for) as 8 gloo ranks sharing this box's GPU(s): a 5-frame clip leaves three ranks without
    a frame -- they still take part in both collectives and return the same result -- and a 19-frame clip gives shares of 3 and 2."""
    import torch.multiprocessing as mp
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), f"r8_{k}.npz")) for k in range(world)]
    m = fv.fvvdp(display_name="standard_fhd")
    for n in (5, 19):
        test, ref = synth_video_pair(n, 54, 96, device="cuda")
        q, st = m.predict(test, ref, frames_per_second=FPS)
        shares = [tuple(int(v) for v in res[k]["share%d" % n]) for k in range(world)]
        assert shares[0][0] == 0 and shares[-1][1] == n and all(a[1] == b[0] for a, b in zip(shares[:-1], shares[1:]))
        if n == 5:
            assert sum(1 for (a, b) in shares if a == b) == 3
        for k in range(world):
            assert np.array_equal(res[k]["Q%d" % n], st["Q_per_ch"]), (n, k)
            assert abs(float(res[k]["jod%d" % n]) - float(q)) < 2e-6


def test_predict_batch_equals_single_calls():
    """fvvdp.predict_batch (BASELINE configs[4]: many independent pairs per GPU): every pair queued on the caller's stream without
    host synchronisation.  Same bits as one synchronous predict() per pair; a failing pair (shape mismatch) raises and leaves the
    metric usable; host-resident, non-contiguous and mixed-dtype pairs (their device copies are temporaries of the caller's
    stream, reused in stream order) give the same results as resident ones."""
    import fovvideovdp_amd as fv
    from fovvideovdp_amd.synth import synth_video_pair
    pairs = [synth_video_pair(24, 135, 240, device="cuda", pair=k) for k in range(5)]
    ref = fv.fvvdp(display_name="standard_fhd")
    want = [ref.predict(t, r, frames_per_second=FPS) for (t, r) in pairs]
    m = fv.fvvdp(display_name="standard_fhd")
    out = m.predict_batch(pairs, frames_per_second=FPS)
    assert len(out) == len(pairs)
    for (q, st), (q0, st0) in zip(out, want):
        assert st["Q_per_ch"].is_cuda
        fv.fvvdp.finish(st)
        assert float(q) == float(q0) and np.array_equal(st["Q_per_ch"], st0["Q_per_ch"])
    # sources the call has to copy or convert first: host arrays, a non-contiguous view, float reference against uint8 test
    odd = [(pairs[0][0].cpu(), pairs[0][1].cpu()),
           (torch.cat([pairs[1][0], pairs[1][0]], 4)[..., :240], pairs[1][1]),
           (pairs[2][0], pairs[2][1].to(torch.float32) / 255)]
    want_odd = [ref.predict(t, r, frames_per_second=FPS) for (t, r) in odd]
    for rep in range(2):
        got = m.predict_batch(odd, frames_per_second=FPS)
        for (q, st), (q0, st0) in zip(got, want_odd):
            fv.fvvdp.finish(st)
            assert float(q) == float(q0) and np.array_equal(st["Q_per_ch"], st0["Q_per_ch"])
    bad = pairs[:2] + [(pairs[0][0], pairs[1][1][:, :, :5])]
    with pytest.raises(RuntimeError):
        m.predict_batch(bad, frames_per_second=FPS)
    q, st = m.predict(pairs[3][0], pairs[3][1], frames_per_second=FPS)          # still usable, synchronous
    assert float(q) == float(want[3][0])
