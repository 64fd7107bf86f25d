#!/usr/bin/env python3
"""Headline benchmark: rollout-steps/sec (encode + N latent steps + decode every step),
NS2d 128x128 3-channel, 64-step rollout, batch 64 per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full `LatentDynamics.predict(x, T=64, to_x=True)` over one batch
(= B*T trajectory-steps).  Inputs are resident in HBM before the timed region; the
timed region is bracketed by barrier + torch.cuda.synchronize() and the MAX over
ranks is taken.  For N > 1 trajectories are sharded over ranks (weak scaling: B per
GPU fixed, no data-path collective) and the decoded shards are all-gathered over
RCCL in step blocks, overlapped with the remaining rollout.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2516.6       # MI355X dense fp16/bf16 matrix peak = 16 x the fp32 matrix rate (same guide)
# The 3x3 convolution runs on the 16-bit matrix pipe with every (scaled) fp32 operand split into two fp16 terms:
# three fp16 products per fp32 product and 10 tap slots for 9 taps = 3.33 executed fp16 FLOP per algorithmic FLOP.
SPLIT_EXEC_PER_ALGO = 3.0 * 10.0 / 9.0
FLOP_PER_TRAJ_STEP = 5.926e9        # 0.732 propagate + 5.194 decode
FLOP_ENCODE = 5.385e9


def build_model(preset, device):
    from helpers import synthetic_state_dict
    from lns_amd import config, dropin
    args = config.preset(preset)
    model = dropin.build_dynamics(args)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synthetic_state_dict(shapes, 1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return args, model.to(device), sd


def cpu_baseline(args, sd, B, T):
    """The CPU oracle (port of the reference path) timed on this box's host cores on a
    bounded sample of the same workload.  Reported, never the thing measured above."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lns_oracle
    from lns_amd import filler
    x = filler.normal("xcpu", (B, args.in_channels, args.Ly, args.Lx), 3)
    # threads: the cores this process may actually use (affinity and cgroup quota), not all
    # hardware threads of the host
    ncpu = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(int(quota) / int(period))))
    except Exception:
        pass
    lns_oracle.set_num_threads(ncpu)
    orc = lns_oracle.OracleDynamics(args, sd)
    orc.predict(x[:1], 1, to_x=True)            # warm the pages / thread pool
    t0 = time.perf_counter()
    orc.predict(x, T, to_x=True)
    dt = time.perf_counter() - t0
    return dict(value=B * T / dt, unit="trajectory-steps/s", cores=lns_oracle.num_threads(), kind="port",
                sample="NS2d 128x128x3, B=%d, T=%d, predict(to_x=True), %.1f s" % (B, T, dt))


def traffic_from_profiles(kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE, separate runs; see profiles/*_traffic.json for the corrections).
    bench.py cannot run the profiler itself; null when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        return {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "unit": "B", "source": os.path.basename(files[-1]),
                "profiled_kernel": k["kernel"], "fetch_bytes_per_launch": k["fetch_bytes_per_launch"],
                "write_bytes_per_launch": k["write_bytes_per_launch"]}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="trajectories per GPU")
    ap.add_argument("--rollout", type=int, default=64, help="latent rollout length T")
    ap.add_argument("--preset", default="ns2d_128")
    ap.add_argument("--gather-chunk", type=int, default=8, help="step-block size of the overlapped all-gather")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) | gloo (rehearsal on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="override the device index (default LOCAL_RANK)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="single-stream execution (no propagate/decode overlap): the mode the per-kernel roofline "
                         "pass uses; profile THIS mode to compare rocprofv3 averages with the roofline block")
    a = ap.parse_args()

    if a.serial:
        os.environ["LNS_NO_OVERLAP"] = "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev_index = a.device if a.device is not None else local_rank
        torch.cuda.set_device(dev_index)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.dist_backend)
    elif a.gpus != 1:
        print("bench.py: --gpus %d needs torch.distributed.run (WORLD_SIZE unset); running 1 GPU" % a.gpus,
              file=sys.stderr)
    n_gpus = world
    dev = torch.device("cuda", a.device if a.device is not None else local_rank)
    torch.cuda.set_device(dev)

    from lns_amd import filler
    args, model, sd = build_model(a.preset, dev)
    B, T = a.batch, a.rollout
    # each rank owns its own contiguous shard of the global batch (rank-dependent seed)
    x = torch.from_numpy(filler.normal("xbench-%d" % rank, (B, args.in_channels, args.Ly, args.Lx), 5)).to(dev)
    eng = model._engine(x)
    # conditional two-phase preset: one normalised parameter per trajectory (U(0,1))
    param = None
    if getattr(args, "family", "") == "twophase_cond":
        param = torch.from_numpy(filler.uniform01("pbench-%d" % rank, B, 5).astype("float32")).to(dev)
    C, H, W = eng.latent_shape()
    out = torch.empty((B, T, args.in_channels, args.Ly, args.Lx), dtype=torch.float32, device=dev)
    gather = world > 1 and not a.no_gather
    from lns_amd import parallel

    def _rollout_latent(z, steps, buf):
        return eng.rollout_latent(z, steps, param=param, to_x=True, out=buf)[1]
    chunked = None
    if gather:
        chunked = parallel.ChunkedGatherRollout(eng.encode, _rollout_latent, (args.in_channels, args.Ly, args.Lx),
                                                B, T, a.gather_chunk, dev, gather=True)

    def one_pass():
        if chunked is None:
            eng.rollout(x, T, param=param, to_x=True, out=out)
        else:
            # chunked rollout; each finished step block is all-gathered on a side stream
            chunked.run(x)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_pass()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    traj_steps = float(n_gpus) * B * T * a.steps
    value = traj_steps / dt

    result = {
        "metric": "rollout-steps/sec (encode+N latent steps+decode), NS2d 128^2 3-ch",
        "value": value, "unit": "trajectory-steps/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seeded N(0,1) fields, deterministic random-init weights)",
        "config": {"workload": "NS2d 128x128 3-channel, %d-step latent rollout, batch=%d per GPU "
                               "(BASELINE.json configs[1])" % (T, B),
                   "preset": a.preset, "batch_per_gpu": B, "rollout_steps": T, "global_batch": B * n_gpus,
                   "parallelism": "trajectory-sharded x%d%s" % (n_gpus, ", overlapped all-gather" if gather else ""),
                   "streams": "single stream" if a.serial else "propagator + 3 decode streams per GPU"},
        "batch_steps_per_s": value / B,
        "path_tflops_per_gpu": (FLOP_PER_TRAJ_STEP * B * T + FLOP_ENCODE * B) * a.steps / dt / 1e12,
    }

    if rank == 0 and not a.no_roofline:
        # per-kernel-class HIP-event timing of one more pass (events recorded around every launch
        # inside the engine, on the stream the kernel is launched on).  The engine runs this
        # diagnostic pass single-stream, so kernel durations are not inflated by co-running
        # kernels; it is kept out of `value`'s timed region.
        eng.timing_enable(True)
        eng.rollout(x, T, param=param, to_x=True, out=out)
        torch.cuda.synchronize()
        tm = eng.timing()
        eng.timing_enable(False)
        k = tm.get("conv3x3_mfma")
        if k:
            ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
            peak = F16_MFMA_PEAK_TFLOPS / SPLIT_EXEC_PER_ALGO
            result["roofline"] = {
                "kernel": "conv3_bf16x3_kernel<..., SPL=2> (3x3 implicit GEMM; fp32 operands as 2 fp16 terms on v_mfma_f32_32x32x16_f16, fp32 accumulate)",
                "bound": "mfma",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "peak_basis": "algorithmic fp32 FLOP: dense fp16 MFMA peak %.1f / %.2f executed fp16 FLOP per algorithmic FLOP"
                              % (F16_MFMA_PEAK_TFLOPS, SPLIT_EXEC_PER_ALGO),
                "executed_f16_tflops": ach * SPLIT_EXEC_PER_ALGO,
                "frac_of_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                # HBM bytes per launch from the committed PMC passes (number, bytes); provenance in traffic_detail
                "traffic": (traffic_from_profiles("conv3x3") or {}).get("hbm_bytes_per_launch"),
                "traffic_detail": traffic_from_profiles("conv3x3"),
                "launches": k["launches"], "avg_launch_us": k["ms"] * 1e3 / k["launches"],
                "algorithmic_flop_per_launch": k["flops"] / k["launches"],
                "algorithmic_bytes_per_launch": k["bytes"] / k["launches"],
                "mode": "single-stream diagnostic pass (python bench.py --serial reproduces it under rocprofv3); the class "
                        "average includes the few stride-2 / thin 3x3 convs that stay on the fp32-MFMA kernel",
                "whole_path_algorithmic_tflops": result["path_tflops_per_gpu"],
            }
        tot = sum(v["ms"] for v in tm.values())
        result["kernel_classes"] = {n: {"ms": round(v["ms"], 3), "share": round(v["ms"] / tot, 4),
                                        "launches": v["launches"],
                                        "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                                        "gbps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                                    for n, v in sorted(tm.items(), key=lambda kv: -kv[1]["ms"])}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, sd, 8, 32)
        result["speedup_vs_cpu_baseline"] = value / result["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
