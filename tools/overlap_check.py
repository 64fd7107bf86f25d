"""Overlapped rollout of a large batch vs (a) the serial single-stream rollout of the same batch and (b) a sub-batch.
All three must agree bitwise.  Repeats to expose timing-dependent faults.

Usage (GPU box):  python tools/overlap_check.py [T] [repeats]
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

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
args = config.preset(os.environ.get("PRESET", "ns2d_128"))
model, _ = gc.build_models(args, 1)
B = int(os.environ.get("BATCH", "64"))
x = filler.normal("xfull", (B, args.in_channels, args.Ly, args.Lx), 5)
xd = torch.from_numpy(x).cuda()
eng = model._engine(xd)
conditional = bool(getattr(model, "_conditional", False))
pd = torch.linspace(0.1, 0.9, B, device="cuda").reshape(B, 1) if conditional else None


def predict(inp, par):
    return model.predict(inp, T, par, to_x=True) if conditional else model.predict(inp, T, to_x=True)


# serial reference: the timing mode runs everything on one stream
eng.timing_enable(True)
ref = predict(xd, pd).clone()
eng.timing_enable(False)
torch.cuda.synchronize()
bad = 0
for r in range(R):
    y = predict(xd, pd)
    torch.cuda.synchronize()
    d = (y - ref).abs().amax(dim=(2, 3, 4))       # [B, T]
    nb = int((d > 0).sum().item())
    bad += nb
    print("repeat %d: mismatching (sample, frame) pairs %d / %d, max %.3e, first bad frame per run %s" % (
        r, nb, d.numel(), d.max().item(), (d > 0).any(0).nonzero().flatten().tolist()[:4]), flush=True)
    if r == 0 and nb:
        idx = (d > 0).nonzero()[:6].tolist()
        for (bi, ti) in idx:
            e = (y[bi, ti] - ref[bi, ti]).abs()
            wrong = e > 0
            rows = wrong.any(0).any(1).nonzero().flatten().tolist()
            cols = wrong.any(0).any(0).nonzero().flatten().tolist()
            print("   sample %d frame %d: %d wrong values, channels %s, rows %s, cols %s" % (
                bi, ti, int(wrong.sum()), wrong.any(2).any(1).nonzero().flatten().tolist(), rows[:12], cols[:40]))
sub = predict(xd[10:12].contiguous(), pd[10:12].contiguous() if conditional else None)
print("sub-batch equals serial big:", torch.equal(sub, ref[10:12]))
print("TOTAL_BAD", bad)
