#!/usr/bin/env python3
"""Host enqueue time vs GPU completion time of one rollout (is the host launch rate the limiter?).
Usage: enqueue_time.py [preset] [B] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from lns_amd import filler
preset = sys.argv[1] if len(sys.argv) > 1 else "ns2d_128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
args, model, sd = bench.build_model(preset, torch.device("cuda", 0))
x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 5)).cuda()
eng = model._engine(x)
out = torch.empty((B, T, args.in_channels, args.Ly, args.Lx), device="cuda")
for _ in range(2):
    eng.rollout(x, T, to_x=True, out=out)
torch.cuda.synchronize()
for it in range(5):
    t0 = time.perf_counter()
    eng.rollout(x, T, to_x=True, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue %.1f ms, complete %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
