#!/usr/bin/env python3
"""Development check of the producer / consumer 3x3 kernel (variants 15 / 16): parity vs oracle and bits vs variant 11."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import gpu_checks as gc

for i, c in enumerate(gc.PC_CASES):
    for v in (15, 16):
        e, shp, y = gc.conv_case(k=3, variant=v, seed=i, ret_y=True, **c)
        _, _, y0 = gc.conv_case(k=3, variant=11, seed=i, ret_y=True, **c)
        print("%-4s v%d %-100s err %.2e  same bits as v11: %s" % ("OK" if e < 2e-6 else "BAD", v, c, e, np.array_equal(y, y0)), flush=True)
