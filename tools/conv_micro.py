#!/usr/bin/env python3
"""Micro-benchmark of single conv layers through lns_op_conv2d (for rocprofv3 --pmc runs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, ctypes
from lns_amd import _lib
import gpu_checks as gc
L = _lib.lib()
cases = {
    "dec13": dict(B=64, Cin=64, Cout=64, H=64, W=64, k=3, up=(128, 128), ss=True, act=1),
    "lat": dict(B=64, Cin=128, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "lat32": dict(B=64, Cin=32, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "lat64": dict(B=64, Cin=64, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "lat256": dict(B=64, Cin=256, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "lat512": dict(B=64, Cin=512, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "c64": dict(B=64, Cin=64, Cout=64, H=64, W=64, k=3, ss=True, act=1),
    "inproj": dict(B=64, Cin=64, Cout=512, H=64, W=64, k=1, ss=True, act=0),
}
which = sys.argv[1:] or list(cases)
VARIANT = int(os.environ.get("CONV_VARIANT", "-1"))   # 6 = bf16x3 kernel
for name in which:
    c = cases[name]
    B, Cin, Cout, H, W, k = c["B"], c["Cin"], c["Cout"], c["H"], c["W"], c["k"]
    Hv, Wv = c.get("up", (H, W))
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = (np.random.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = np.zeros(Cout, np.float32)
    ss = torch.stack([1 + 0.1 * torch.randn(B, Cin), 0.1 * torch.randn(B, Cin)], -1).cuda().contiguous()
    p = (k - 1) // 2
    y = torch.empty(B, Cout, Hv, Wv, device="cuda")
    for it in range(3):
        rc = L.lns_op_conv2d(x.data_ptr(), B, Cin, H, W, Hv, Wv, gc._hp(w), gc._hp(bias), Cout, k, 1, 1, p, p, p, p, 1, 1,
                             ss.data_ptr(), c["act"], 0, None, None, y.data_ptr(), VARIANT if k == 3 else -1, None, None)
        assert rc == 0
    torch.cuda.synchronize()
    print(name, "done")
