"""Layer-by-layer comparison of a large-batch run with a sub-batch run (must be bit-identical).

Usage (GPU box):  python tools/batch_trace_diff.py [preset] [B]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import gpu_checks as gc  # noqa: E402
from lns_amd import config, filler  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "ns2d_128"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
args = config.preset(preset)
model, _ = gc.build_models(args, 1)
x = filler.normal("xfull", (B, args.in_channels, args.Ly, args.Lx), 5)
xd = torch.from_numpy(x).cuda()
sl = slice(10, 12)


def traces(xin):
    eng = model._engine(xin)
    out = {}
    eng.trace_enable(True)
    z = eng.encode(xin)
    torch.cuda.synchronize()
    out["enc"] = list(eng.trace())
    eng.trace_enable(True)
    z1 = eng.propagate(z)
    torch.cuda.synchronize()
    out["prop"] = list(eng.trace())
    eng.trace_enable(True)
    y = eng.decode(z1)
    torch.cuda.synchronize()
    out["dec"] = list(eng.trace())
    eng.trace_enable(False)
    return out, y


big, yb = traces(xd)
small, ys = traces(xd[sl].contiguous())
for stage in ("enc", "prop", "dec"):
    for (n1, a), (n2, b) in zip(big[stage], small[stage]):
        assert n1 == n2
        d = np.abs(a[sl].astype(np.float64) - b).max()
        flag = "" if d == 0 else "   <-- DIFF"
        print("%-5s %-50s %s maxdiff %.3e%s" % (stage, n1, a.shape, d, flag), flush=True)
print("final equal:", torch.equal(yb[sl], ys))
