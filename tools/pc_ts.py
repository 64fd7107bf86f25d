#!/usr/bin/env python3
"""Who waits for whom in the producer / consumer 3x3 kernel?  Needs a -DLNS_PC_TS build (tools/build_variant.sh pcts
-DLNS_PC_TS) loaded through LNS_HIP_LIB.  Runs one layer once and prints, per role, the shader-clock cycles between
leaving a barrier and arriving at the next one ("busy") and the cycles spent inside the barrier ("wait")."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from lns_amd import _lib
import gpu_checks as gc

L = _lib.lib()
N = 64
case = dict(B=64, Cin=128, Cout=128, H=64, W=64)
if len(sys.argv) > 2:
    case = dict(B=int(sys.argv[2]), Cin=int(sys.argv[3]), Cout=int(sys.argv[4]), H=int(sys.argv[5]), W=int(sys.argv[6]))
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, Cin, Cout, H, W = case["B"], case["Cin"], case["Cout"], case["H"], case["W"]
x = torch.randn(B, Cin, H, W, device="cuda")
w = (np.random.RandomState(0).randn(Cout, Cin, 3, 3) / np.sqrt(Cin * 9)).astype(np.float32)
ss = torch.stack([1 + 0.1 * torch.randn(B, Cin), 0.1 * torch.randn(B, Cin)], -1).cuda().contiguous()
y = torch.empty(B, Cout, H, W, device="cuda")
for rep in range(2):
    rc = L.lns_op_conv2d(x.data_ptr(), B, Cin, H, W, H, W, gc._hp(w), None, Cout, 3, 1, 1, 1, 1, 1, 1, 1, 1, ss.data_ptr(), 1, 0,
                         None, None, y.data_ptr(), variant, None, None)
    assert rc == 0
torch.cuda.synchronize()
buf = np.zeros(512 * 2 * N * 2, np.int64)
f = L.lns_debug_pc_ts
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert f(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
t = buf.reshape(512, 2, N, 2)
live = t[:, 0, 8, 0] != 0
t = t[live]
print("variant %d case %s: %d blocks recorded" % (variant, case, t.shape[0]))
for role, name in ((0, "consumer"), (1, "producer")):
    arrive, leave = t[:, role, :, 0], t[:, role, :, 1]
    busy = arrive[:, 1:] - leave[:, :-1]
    wait = leave - arrive
    it = slice(8, N - 1)
    print("  %-9s busy mean %7.0f  median %7.0f  p90 %7.0f cycles | wait in barrier mean %7.0f median %7.0f" % (
        name, busy[:, it].mean(), np.median(busy[:, it]), np.percentile(busy[:, it], 90), wait[:, 9:].mean(), np.median(wait[:, 9:])))
period = (t[:, 0, 40, 1] - t[:, 0, 8, 1]) / 32.0
print("  iteration period (consumer leave-to-leave): mean %.0f cycles" % period.mean())
b0 = t[0]
print("  block 0, iterations 8..40: consumer busy / producer busy")
print("   ", " ".join("%d/%d" % (b0[0, i + 1, 0] - b0[0, i, 1], b0[1, i + 1, 0] - b0[1, i, 1]) for i in range(8, 40)))

if hasattr(L, "lns_debug_pc_ts2"):
    b2 = np.zeros(512 * N * 4, np.int64)
    f2 = L.lns_debug_pc_ts2
    f2.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert f2(b2.ctypes.data_as(ctypes.c_void_p), b2.size) == 0
    u = b2.reshape(512, N, 4)[live]
    it = slice(8, N - 1)
    leave = t[:, 1, :, 1]
    print("  producer phases (cycles, median over blocks x iterations 8..62): side jobs %d | restage (transform + LDS writes + loads) %d | barrier arrive %d" % (
        np.median(u[:, it, 1] - u[:, it, 0]), np.median(u[:, it, 2] - u[:, it, 1]), np.median(t[:, 1, it, 0] - u[:, it, 2])))
    print("  producer restage per iteration, block 0, iterations 8..40:", " ".join(str(int(v)) for v in (u[0, 8:40, 2] - u[0, 8:40, 1])))
    print("  producer side jobs per iteration, block 0, iterations 8..40:", " ".join(str(int(v)) for v in (u[0, 8:40, 1] - u[0, 8:40, 0])))
