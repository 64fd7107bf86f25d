"""Accuracy of the bf16x3 3x3 conv kernel vs the fp32-MFMA kernel, both against an fp64 convolution.

Usage (GPU box):  python tools/bf16x3_check.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gpu_checks as G  # noqa: E402
from lns_amd import _lib  # noqa: E402


def run(B, Cin, Cout, H, W, dil, mode, variant, scale=1.0, seed=0):
    L = _lib.lib()
    r = np.random.default_rng(seed)
    x = (r.standard_normal((B, Cin, H, W)) * scale).astype(np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    xt = torch.from_numpy(x).double()
    pm = ["zeros", "circular"]
    xp = torch.nn.functional.pad(xt, (dil, dil, 0, 0), mode="circular" if mode[1] else "constant")
    xp = torch.nn.functional.pad(xp, (0, 0, dil, dil), mode="circular" if mode[0] else "constant")
    ref = torch.nn.functional.conv2d(xp, torch.from_numpy(w).double(), dilation=dil).numpy()
    xd = torch.from_numpy(x).cuda()
    y = torch.full(ref.shape, float("nan"), dtype=torch.float32, device="cuda")
    rc = L.lns_op_conv2d(xd.data_ptr(), B, Cin, H, W, H, W, G._hp(w), None, Cout, 3, 1, dil, dil, dil, dil, dil,
                         mode[0], mode[1], None, 0, 0, None, None, y.data_ptr(), variant, G._stream(), None)
    assert rc == 0, rc
    out = y.cpu().numpy().astype(np.float64)
    assert np.isfinite(out).all()
    err = out - ref
    return float(np.sqrt((err ** 2).sum() / (ref ** 2).sum())), float(np.abs(err).max() / np.abs(ref).max())


if __name__ == "__main__":
    cases = [(2, 64, 64, 32, 32, 1, (1, 1)), (2, 128, 128, 16, 16, 2, (1, 1)), (2, 512, 128, 32, 32, 1, (0, 0)),
             (2, 64, 64, 24, 48, 1, (0, 1)), (2, 128, 128, 12, 24, 3, (0, 1)), (1, 64, 64, 128, 128, 1, (1, 1)),
             (2, 40, 72, 20, 36, 1, (0, 0))]
    for c in cases:
        for scale in (1.0, 1e-3, 100.0):
            e1 = run(*c, variant=1, scale=scale)
            e6 = run(*c, variant=6, scale=scale)
            e11 = run(*c, variant=11, scale=scale)
            print("case %s scale %g: fp32-mfma rel %.3e max %.3e | bf16x3 rel %.3e max %.3e | f16x2 rel %.3e max %.3e" % (
                c, scale, e1[0], e1[1], e6[0], e6[1], e11[0], e11[1]), flush=True)
