#!/usr/bin/env python3
"""Quick parity of conv variant 14 (256-pixel tiles) against variant 11 and the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gpu_checks as gc
for kw in (dict(B=2, Cin=64, Cout=64, H=64, W=64, ss=True, act_in=1),
           dict(B=2, Cin=40, Cout=100, H=32, W=32, ss=True, act_in=1, res=True, badd=True),
           dict(B=3, Cin=16, Cout=64, H=24, W=40, ss=False),
           dict(B=2, Cin=64, Cout=64, H=16, W=16, up=(32, 32), ss=True, act_in=1),
           dict(B=2, Cin=32, Cout=128, H=16, W=16, dil=2, ss=True, act_in=2, act_out=2),
           dict(B=2, Cin=24, Cout=64, H=61, W=121, ss=True, act_in=1, mode=(0, 0))):
    for v in (11, 14):
        r = gc.conv_case(variant=v, **kw)
        print(v, kw, "err %.3e" % r[0], r[1])
