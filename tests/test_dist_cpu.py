"""CPU, gloo, world_size 2: the N>1 orchestration (trajectory sharding + chunked overlapped
all-gather of lns_amd.parallel) reproduces the unsharded rollout exactly.  The per-rank compute is
the CPU oracle here (test infrastructure) -- on GPUs the same orchestration drives the HIP engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, chunk, q, mode="chunked"):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lns_oracle
    from helpers import manifest, synthetic_state_dict
    from lns_amd import config, filler, parallel
    lns_oracle.set_num_threads(2)
    args = config.preset("ns2d_mini")
    sd = synthetic_state_dict({k: tuple(v) for k, v in manifest()["ns2d_mini"].items()}, 1)
    orc = lns_oracle.OracleDynamics(args, sd)
    GB = 4
    x_all = filler.normal("xdist", (GB, args.in_channels, args.Ly, args.Lx), 9)
    lo, hi = parallel.shard_bounds(GB, rank, world)

    def encode(x):
        return torch.from_numpy(orc.x_to_z(x.numpy()))

    def rollout_latent(z, steps, out):
        zz = z.numpy()
        for t in range(steps):
            zz = orc.prop.forward(zz)
            out[:, t] = torch.from_numpy(orc.z_to_x(zz))
        return torch.from_numpy(zz)

    if mode == "end":       # the north star's form: whole rollout, then ONE all-gather into [world*B, T, C, H, W]
        def rollout(x, out):
            rollout_latent(encode(x), T, out)
        r = parallel.EndGatherRollout(rollout, (args.in_channels, args.Ly, args.Lx), hi - lo, T, "cpu")
    else:
        r = parallel.ChunkedGatherRollout(encode, rollout_latent, (args.in_channels, args.Ly, args.Lx), hi - lo, T,
                                          chunk, "cpu")
    r.run(torch.from_numpy(x_all[lo:hi]))
    full = r.assemble().numpy()
    if rank == 0:
        ref = orc.predict(x_all, T, to_x=True)
        q.put((full.shape, float(np.abs(full - ref).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rollout_with_overlapped_gather_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 9, 2, q)) for r in range(2)]   # blocks [4, 2, 2, 1]
    for p in procs:
        p.start()
    shape, err = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert shape == (4, 9, 2, 32, 32)
    assert err == 0.0          # trajectories are independent: sharded == unsharded bit for bit


def test_sharded_rollout_with_single_end_gather_matches_unsharded():
    """--gather-mode end: one all_gather_into_tensor of the decoded shards after the rollout (gloo, world_size 2)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, 2, q, "end")) for r in range(2)]
    for p in procs:
        p.start()
    shape, err = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert shape == (4, 5, 2, 32, 32)
    assert err == 0.0


def _metric_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lns_oracle
    from helpers import METRIC_STATS, metric_inputs
    from lns_amd import parallel
    st = METRIC_STATS["ns2d"]
    yh, y = metric_inputs("ns2d")                               # global batch of 2: one trajectory per rank
    lo, hi = parallel.shard_bounds(yh.shape[0], rank, world)
    f, s = lns_oracle.rollout_metrics(yh[lo:hi], y[lo:hi], st["mean"], st["std"])   # per-rank reduction (oracle on CPU)
    f_all, s_all = parallel.gather_metrics(torch.from_numpy(f), torch.from_numpy(s))
    if rank == 0:
        f_ref, s_ref = lns_oracle.rollout_metrics(yh, y, st["mean"], st["std"])
        q.put((tuple(f_all.shape), tuple(s_all.shape), bool((f_all.numpy() == f_ref).all()),
               bool((s_all.numpy() == s_ref).all())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_metrics_gather_only_the_reductions():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metric_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    fshape, sshape, f_ok, s_ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert fshape == (2, 5, 3) and sshape == (2, 3)
    assert f_ok and s_ok       # per-trajectory reductions: sharded == unsharded


def test_shard_bounds_cover_the_batch():
    from lns_amd import parallel
    for gb in (1, 7, 64, 512):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(gb, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    assert parallel.chunk_lengths(64, 16) == [32, 16, 16]
    assert parallel.chunk_lengths(64, 8) == [32, 16, 8, 8]
    assert parallel.chunk_lengths(256, 8) == [128, 64, 32, 16, 8, 8]
    assert sum(parallel.chunk_lengths(37, 4)) == 37
    assert parallel.chunk_lengths(5, 2) == [2, 2, 1]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment: the parent makes no GPU call, starts two ranks
    through torch.distributed.run, relays rank 0's JSON line and exits with the children's status.  CPU rehearsal:
    gloo backend and --plumbing-only (sharding + step-block all-gather + barrier / max-over-ranks timing, no compute)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--plumbing-only", "--steps", "2", "--warmup", "1", "--batch", "3", "--rollout", "9",
                        "--gather-chunk", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]           # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["plumbing_ok"] is True
    assert rec["config"]["global_batch"] == 6 and rec["steps"] == 2 and rec["warmup"] == 1
    assert "one end-of-rollout all-gather" in rec["config"]["parallelism"]          # the default mode
    mg = rec["multi_gpu"]                                    # self-describing: what the backend saw, what moved
    assert mg["backend"] == "gloo" and mg["world_size_seen_by_backend"] == 2 and mg["gather"] == "end"
    assert mg["gathers_per_rollout"] == 1 and mg["bytes_contributed_per_rank"] == 4 * 3 * 9 * 2 * 8 * 8
    assert mg["bytes_received_per_rank"] == mg["bytes_contributed_per_rank"] and mg["exposed_gather_ms_rank0"] >= 0.0
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--plumbing-only", "--steps", "1", "--warmup", "0", "--batch", "3", "--rollout", "9",
                        "--gather-chunk", "2", "--gather-mode", "chunked"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert "overlapped step-block all-gather" in rec["config"]["parallelism"] and rec["plumbing_ok"] is True
    assert rec["multi_gpu"]["gather"] == "chunked" and rec["multi_gpu"]["gathers_per_rollout"] == 4


def test_bench_force_dist_runs_the_collective_with_one_rank():
    """`python bench.py --gpus 1 --force-dist`: the process group is created IN this process with ONE rank and the
    end-of-rollout all_gather_into_tensor runs anyway, into a separate [world*B, T, C, H, W] buffer that must equal the shard
    bit for bit.  CPU rehearsal of the RCCL readiness run (gloo, --plumbing-only); tests/test_gpu_parity.py runs it on RCCL."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "gloo",
                        "--plumbing-only", "--steps", "2", "--warmup", "1", "--batch", "3", "--rollout", "5"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    mg = rec["multi_gpu"]
    assert rec["n_gpus"] == 1 and rec["plumbing_ok"] is True
    assert mg["backend"] == "gloo" and mg["world_size_seen_by_backend"] == 1 and mg["force_dist"] is True
    assert mg["gather"] == "end" and mg["gathers_per_rollout"] == 1 and mg["gathered_equals_shard"] is True
    assert mg["bytes_contributed_per_rank"] == 4 * 3 * 5 * 2 * 8 * 8 and mg["bytes_received_per_rank"] == 0


def test_bench_workload_labels_follow_the_preset():
    """metric / workload strings are derived from the preset (they were hard-coded to NS2d in round 1)."""
    sys.path.insert(0, ROOT)
    import bench
    for preset, (label, wl, fixtures) in bench.WORKLOADS.items():
        assert label and wl
        for T, fx in fixtures.items():
            assert os.path.exists(os.path.join(ROOT, "tests", "golden", fx + ".npz")), fx
    for (preset, T), fx in bench.STABLE_FIXTURES.items():
        assert preset in bench.WORKLOADS and os.path.exists(os.path.join(ROOT, "tests", "golden", fx + ".npz")), fx
    assert "NS2d" in bench.WORKLOADS["ns2d_128"][0] and "two-phase" in bench.WORKLOADS["twophase_cond"][0]
