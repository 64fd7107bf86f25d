#!/usr/bin/env python3
"""Same-process A/B of the rollout's scheduling options (lns_set_option): interleaved rounds, median and min per arm.
    python tools/sched_sweep.py [preset] [B] [T] [rounds]
Every arm produces bit-identical fields (checked on the first round)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from lns_amd import filler

preset = sys.argv[1] if len(sys.argv) > 1 else "ns2d_128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
R = int(sys.argv[4]) if len(sys.argv) > 4 else 5
GROUPS = [int(v) for v in os.environ.get("SWEEP_GROUPS", "1,2,4").split(",")]
STREAMS = [int(v) for v in os.environ.get("SWEEP_STREAMS", "2,3").split(",")]
PRIOS = [int(v) for v in os.environ.get("SWEEP_PRIO", "0").split(",")]
CHUNKS = [int(v) for v in os.environ.get("SWEEP_FA_CHUNK_MB", "0").split(",")]      # FABlock chain per group of samples (fa_chunk_mb)
ARMS = [dict(decode_group=g, decode_streams=s, prop_priority=p, fa_chunk_mb=c) for g in GROUPS for s in STREAMS for p in PRIOS for c in CHUNKS]
if os.environ.get("SWEEP_SERIAL"):
    ARMS = [dict(a, overlap=0) for a in ARMS]
args, model, sd = bench.build_model(preset, torch.device("cuda", 0))
x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 5)).cuda()
param = torch.from_numpy(filler.uniform01("p", B, 5).astype("float32")).cuda() if args.family == "twophase_cond" else None
eng = model._engine(x)
out = torch.empty((B, T, args.in_channels, args.Ly, args.Lx), dtype=torch.float32, device="cuda")
ref = None
times = {i: [] for i in range(len(ARMS))}
for r in range(R + 1):
    for i, arm in enumerate(ARMS):
        for k, v in arm.items():
            eng.set_option(k, v)
        print("round %d arm %s" % (r, arm), file=sys.stderr, flush=True)
        eng.rollout(x, T, param=param, to_x=True, out=out)        # (first call after an option change allocates)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout(x, T, param=param, to_x=True, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r == 0:
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), ("arm changes the result", arm)
        else:
            times[i].append(dt)
for i, arm in enumerate(ARMS):
    t = sorted(times[i])
    print(json.dumps(dict(arm=arm, median_ms=round(1e3 * t[len(t) // 2], 2), min_ms=round(1e3 * t[0], 2),
                          traj_steps_per_s=round(B * T / t[len(t) // 2]))))
