#!/usr/bin/env python3
"""Rollout time with and without the decode (the latent chain alone): python tools/chain_only.py [preset] [B] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from lns_amd import filler
preset = sys.argv[1] if len(sys.argv) > 1 else "twophase_cond"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T = int(sys.argv[3]) if len(sys.argv) > 3 else 128
args, model, sd = bench.build_model(preset, torch.device("cuda", 0))
x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 5)).cuda()
param = torch.from_numpy(filler.uniform01("p", B, 5).astype(np.float32)).cuda() if args.family == "twophase_cond" else None
eng = model._engine(x)
for to_x in (True, False, True, False):
    eng.rollout(x, T, param=param, to_x=to_x); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.rollout(x, T, param=param, to_x=to_x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("to_x=%s: %.2f ms" % (to_x, min(ts) * 1e3))
# the decode alone: T decodes of B latents, back to back on one stream
z = model.x_to_z(x, param) if param is not None and getattr(args, "cond_encoder", False) else model.x_to_z(x)
model.z_to_x(z); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(T):
        model.z_to_x(z)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("decode alone, %d x B=%d: %.2f ms" % (T, B, min(ts) * 1e3))
