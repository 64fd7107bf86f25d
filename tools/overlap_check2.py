"""Four-way consistency: {overlapped, single-stream} x {full batch, sub-batch}, plus the oracle on the sub-batch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import gpu_checks as gc  # noqa: E402
from helpers import rel_l2  # noqa: E402
from lns_amd import config, filler  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
args = config.preset("ns2d_128")
model, orc = gc.build_models(args, 1)
x = filler.normal("xfull", (64, args.in_channels, args.Ly, args.Lx), 5)
xd = torch.from_numpy(x).cuda()
xs = xd[10:12].contiguous()
eng = model._engine(xd)


def serial(inp):
    eng.timing_enable(True)
    r = model.predict(inp, T, to_x=True).clone()
    eng.timing_enable(False)
    torch.cuda.synchronize()
    return r


res = {}
for rep in range(2):
    res["ser_big%d" % rep] = serial(xd)[10:12].clone()
    res["ov_big%d" % rep] = model.predict(xd, T, to_x=True)[10:12].clone()
    res["ser_sub%d" % rep] = serial(xs)
    res["ov_sub%d" % rep] = model.predict(xs, T, to_x=True).clone()
torch.cuda.synchronize()
keys = list(res)
ref = orc.predict(x[10:12], T, to_x=True)
for k in keys:
    print("%-10s vs oracle rel_l2 %.3e   per-frame maxdiff vs ser_sub0: %s" % (
        k, rel_l2(res[k].cpu().numpy(), ref),
        ["%.1e" % (res[k][:, t] - res["ser_sub0"][:, t]).abs().max().item() for t in range(T)]), flush=True)
