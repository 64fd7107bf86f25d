#!/usr/bin/env python3
"""Cost of the epilogue activation of one latent-resolution 3x3 conv (run under rocprofv3 --kernel-trace --stats).
Usage: act_cost.py <act_out: 0 none | 1 swish | 2 gelu>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from lns_amd import _lib
import gpu_checks as gc
L = _lib.lib()
act_out = int(sys.argv[1])
B, Cin, Cout, H, W, k = 64, 128, 128, 16, 16, 3
x = torch.randn(B, Cin, H, W, device="cuda")
w = (np.random.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
bias = np.zeros(Cout, np.float32)
ss = torch.stack([1 + 0.1 * torch.randn(B, Cin), 0.1 * torch.randn(B, Cin)], -1).cuda().contiguous()
y = torch.empty(B, Cout, H, W, device="cuda")
for it in range(20):
    rc = L.lns_op_conv2d(x.data_ptr(), B, Cin, H, W, H, W, gc._hp(w), gc._hp(bias), Cout, k, 1, 1, 1, 1, 1, 1, 1, 1,
                         ss.data_ptr(), 0, act_out, None, None, y.data_ptr(), 11, None, None)
    assert rc == 0
torch.cuda.synchronize()
print("done", act_out)
