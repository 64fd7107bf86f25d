#!/usr/bin/env python3
"""In-kernel clock of the split-operand conv kernels' K loop from a -DLNS_TS=3 build's stamps ($LNS_TS_FILE): per block the
100 MHz wall clock (s_memrealtime) and the shader clock counter (s_memtime) at loop start and loop end.

    clock = (shader ticks) / (wall ticks) x 100 MHz        (MI355X_MICROARCH.md, DVFS give-back item 6)

    python tools/clock_analyze.py ts.txt
"""
import sys
import numpy as np

launches, cur, hdr = [], [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur:
            launches.append((hdr, np.array(cur, dtype=np.int64)))
        hdr, cur = line.strip(), []
    else:
        cur.append([int(v) for v in line.split()])
if cur:
    launches.append((hdr, np.array(cur, dtype=np.int64)))
by = {}
for hdr, a in launches:
    a = a[(a[:, 2] > a[:, 1]) & (a[:, 4] > a[:, 3])]          # columns: block, wall0, wall1, shader0, shader1, ...
    if not len(a):
        continue
    wall = (a[:, 2] - a[:, 1]).astype(np.float64)             # 10 ns ticks
    shader = (a[:, 4] - a[:, 3]).astype(np.float64)
    ok = wall >= 50                                           # loops of at least 0.5 us (tick resolution)
    if ok.sum() == 0:
        continue
    ghz = shader[ok] / wall[ok] * 0.1
    by.setdefault(hdr.split(" B=")[0] + " B=" + hdr.split(" B=")[1], []).append((np.median(ghz), np.percentile(ghz, 10), np.percentile(ghz, 90),
                                                                               np.median(wall[ok]) / 100.0, int(ok.sum())))
for hdr, v in by.items():
    v = np.array(v)
    print("%s\n   launches stamped %d, blocks per launch %d: K loop %.2f us (median); in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f)" % (
        hdr, len(v), int(v[:, 4].mean()), np.median(v[:, 3]), np.median(v[:, 0]), np.median(v[:, 1]), np.median(v[:, 2])))
