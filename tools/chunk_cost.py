#!/usr/bin/env python3
"""Cost of running the rollout as step-blocks (the multi-GPU path gathers finished blocks while the next one runs):
one 64-step call vs encode + k calls of 64/k steps through rollout_latent.  Usage: chunk_cost.py [B] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from lns_amd import filler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
args, model, sd = bench.build_model("ns2d_128", torch.device("cuda", 0))
x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 5)).cuda()
eng = model._engine(x)
out = torch.empty((B, T, args.in_channels, args.Ly, args.Lx), device="cuda")


def timed(fn, n=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("one call: %.1f ms" % timed(lambda: eng.rollout(x, T, to_x=True, out=out)), flush=True)
for chunk in (32, 16, 8, 4):
    bufs = [torch.empty((B, chunk, args.in_channels, args.Ly, args.Lx), device="cuda") for _ in range(T // chunk)]

    def run():
        z = eng.encode(x)
        for b in bufs:
            z = eng.rollout_latent(z, chunk, to_x=True, out=b)[1]
    print("blocks of %2d: %.1f ms" % (chunk, timed(run)), flush=True)
