"""Co-residency stress: two convolutions on two streams, each checked bitwise against its solo result.

Usage (GPU box):  python tools/pair_stress.py
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (initialises the HIP runtime the same way the engine's users do)
from lns_amd import _lib  # noqa: E402

L = _lib.lib()
L.lns_op_conv_pair_stress.restype = ctypes.c_int
L.lns_op_conv_pair_stress.argtypes = [ctypes.c_int] * 13 + [ctypes.POINTER(ctypes.c_longlong)] * 2


def run(name, B, H, W, a, b, rounds=6, launches=6):
    ma, mb = ctypes.c_longlong(0), ctypes.c_longlong(0)
    rc = L.lns_op_conv_pair_stress(B, H, W, a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], rounds, launches,
                                   ctypes.byref(ma), ctypes.byref(mb))
    print("%-46s rc=%d  mismatching words: A %d  B %d" % (name, rc, ma.value, mb.value), flush=True)


if __name__ == "__main__":
    torch.cuda.init()
    # (Cin, Cout, ksize, variant)
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        run("bf16x3 3x3 64->64  | fp32 1x1 64->64", 32, 64, 64, (64, 64, 3, 6), (64, 64, 1, -1), rounds=12)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "exp":       # the pairs that failed on round 2's packed-fp32 build, once each
        run("bf16x3 1x1 64->64  | fp32 1x1 64->64", 32, 64, 64, (64, 64, 1, 7), (64, 64, 1, 1), rounds=12)
        run("bf16x3 3x3 64->64  | fp32 1x1 64->512", 32, 64, 64, (64, 64, 3, 6), (64, 512, 1, -1), rounds=12)
        run("bf16x3 3x3 64->64  | fp32 1x1 64->64 variant 4", 32, 64, 64, (64, 64, 3, 6), (64, 64, 1, 4))
        run("bf16x3 3x3 128->128 16x16 | fp32 1x1 128->128", 64, 16, 16, (128, 128, 3, 6), (128, 128, 1, -1))
        sys.exit(0)
    run("bf16x3 3x3 64->64  | fp32 1x1 64->64", 32, 64, 64, (64, 64, 3, 6), (64, 64, 1, -1))
    run("bf16x3 3x3 128->128 16x16 | fp32 1x1 128->128", 64, 16, 16, (128, 128, 3, 6), (128, 128, 1, -1))
    for v in (1, 3, 4):
        run("bf16x3 3x3 64->64  | fp32 1x1 64->64 variant %d" % v, 32, 64, 64, (64, 64, 3, 6), (64, 64, 1, v))
    run("bf16x3 3x3 64->64  | fp32 1x1 64->512", 32, 64, 64, (64, 64, 3, 6), (64, 512, 1, -1), rounds=12)
    run("fp32 3x3 64->64    | fp32 1x1 64->64", 32, 64, 64, (64, 64, 3, 1), (64, 64, 1, -1), rounds=12)
    run("fp32 1x1 64->64    | fp32 1x1 64->64", 32, 64, 64, (64, 64, 1, -1), (64, 64, 1, -1), rounds=12)
    run("bf16x3 3x3 64->64  | bf16x3 1x1 64->64", 32, 64, 64, (64, 64, 3, 6), (64, 64, 1, 7), rounds=12)
    run("bf16x3 1x1 64->512 | bf16x3 1x1 512->64", 16, 64, 64, (64, 512, 1, 7), (512, 64, 1, 7), rounds=12)
    run("bf16x3 1x1 64->64  | fp32 1x1 64->64", 32, 64, 64, (64, 64, 1, 7), (64, 64, 1, 1), rounds=12)
