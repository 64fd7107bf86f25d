#!/usr/bin/env python3
"""Per-layer HIP-event timing of one rollout (LNS_TIMING_BY_NAME=1). Usage: layer_times.py [preset] [B] [T]"""
import os, sys
os.environ["LNS_TIMING_BY_NAME"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from lns_amd import filler
preset = sys.argv[1] if len(sys.argv) > 1 else "ns2d_128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
args, model, sd = bench.build_model(preset, torch.device("cuda", 0))
x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 5)).cuda()
import numpy as np
param = torch.from_numpy(filler.uniform01("p", B, 5).astype(np.float32)).cuda() if args.family == "twophase_cond" else None
eng = model._engine(x)
eng.rollout(x, 2, param=param, to_x=True)
torch.cuda.synchronize()
eng.timing_enable(True)
eng.rollout(x, T, param=param, to_x=True)
torch.cuda.synchronize()
tm = eng.timing()
tot = sum(v["ms"] for v in tm.values())
print("total %.2f ms for T=%d (%.2f ms/step)" % (tot, T, tot / T))
for n, v in sorted(tm.items(), key=lambda kv: -kv[1]["ms"]):
    per = v["ms"] / v["launches"]
    print("%-70s %8.3f ms  %5.1f%%  n=%4d  %8.1f us/launch  %6.1f TF/s  %7.1f GB/s" % (
        n, v["ms"], 100 * v["ms"] / tot, v["launches"], per * 1e3,
        v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0, v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["bytes"] else 0))
