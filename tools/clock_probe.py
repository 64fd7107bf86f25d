#!/usr/bin/env python3
"""The clock the chip holds inside the split-operand conv kernels' K loops during a sustained single-stream rollout
(needs a -DLNS_TS=3 build: tools/build_variant.sh ts3 -DLNS_TS=3):

    LNS_HIP_LIB=<ts3 lib> LNS_TS_FILE=ts.txt LNS_TS_LAYER=<op name substring> LNS_TS_SKIP=<launches> LNS_TS_MAX=3 \
        python tools/clock_probe.py [preset] [B] [rollouts]
    python tools/clock_analyze.py ts.txt
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from lns_amd import filler
preset = sys.argv[1] if len(sys.argv) > 1 else "ns2d_128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
R = int(sys.argv[3]) if len(sys.argv) > 3 else 24
args, model, sd = bench.build_model(preset, torch.device("cuda", 0))
x = torch.from_numpy(filler.normal("x", (B, args.in_channels, args.Ly, args.Lx), 5)).cuda()
eng = model._engine(x)
eng.set_option("overlap", 0)
out = torch.empty((B, 64, args.in_channels, args.Ly, args.Lx), dtype=torch.float32, device="cuda")
t0 = time.time()
for _ in range(R):
    eng.rollout(x, 64, to_x=True, out=out)
torch.cuda.synchronize()
print("done: %d rollouts in %.1f s" % (R, time.time() - t0))
