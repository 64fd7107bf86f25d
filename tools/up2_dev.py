#!/usr/bin/env python3
"""Development check of the phase-decomposed upsample conv (variant 17) vs the oracle and vs the nine-tap gather form (11)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import gpu_checks as gc
from helpers import rel_l2
for i, c in enumerate(gc.UP2_CASES):
    e, shp, y = gc.conv_case(k=3, variant=17, seed=i, ret_y=True, **c)
    _, _, y0 = gc.conv_case(k=3, variant=11, seed=i, ret_y=True, **c)
    print("%-4s %-110s err %.2e  vs nine-tap form %.2e" % ("OK" if e < 2e-6 else "BAD", c, e, rel_l2(y, y0)), flush=True)
