#!/usr/bin/env python3
"""Per-block phase timestamps of one layer inside a rollout (needs a -DLNS_TS build of the library):
    LNS_HIP_LIB=build/lns_ts.so LNS_TS_FILE=ts.txt LNS_TS_LAYER=model.11.to_out python tools/ts_layer.py [preset] [B]
then  python tools/ts_analyze.py ts.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from lns_amd import filler
preset = sys.argv[1] if len(sys.argv) > 1 else "ns2d_128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
args, model, sd = bench.build_model(preset, torch.device("cuda", 0))
x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 5)).cuda()
eng = model._engine(x)
eng.set_option("overlap", 0)
eng.rollout(x, 2, to_x=True)
torch.cuda.synchronize()
print("done")
