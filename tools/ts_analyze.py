#!/usr/bin/env python3
"""Summarise the per-block phase timestamps a -DLNS_TS build of liblns_hip.so appends to $LNS_TS_FILE
(lns_op_conv2d, split-operand 3x3 kernel).  Timestamps are 100 MHz wall-clock ticks.

    LNS_HIP_LIB=build/lns_ts.so LNS_TS_FILE=ts.txt python tools/conv_time.py c64 ; python tools/ts_analyze.py ts.txt
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
seen = {}
for hdr, a in launches:
    seen[hdr] = a                      # last launch of each shape (warm)
for hdr, a in seen.items():
    a = a[a[:, 1] != 0]                 # rows of blocks that do not exist in this launch form
    ts = a[:, 1:7].astype(np.float64) / 100.0          # us
    hw, xcc = a[:, 7], a[:, 8]
    cu = ((xcc & 0xF) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    t0 = ts[:, 0].min()
    print(hdr)
    print("  kernel span (first entry -> last store ack): %.2f us; distinct CUs %d" % (ts[:, 5].max() - t0, len(set(cu.tolist()))))
    names = ["entry->tables published", "->stage 0 in LDS (loop start)", "main loop", "epilogue (stores issued)", "store ack + amax"]
    for i, n in enumerate(names):
        d = ts[:, i + 1] - ts[:, i]
        print("  %-32s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (n, d.mean(), *np.percentile(d, [10, 50, 90])))
    life = ts[:, 5] - ts[:, 0]
    print("  %-32s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % ("block lifetime", life.mean(), *np.percentile(life, [10, 50, 90])))
    # per CU: fraction of the kernel span with >= 1 block inside its main loop, and mean number of blocks in the loop
    span = ts[:, 5].max() - t0
    in_loop, occ = [], []
    for c in set(cu.tolist()):
        m = cu == c
        ev = sorted([(t, 1) for t in ts[m, 2]] + [(t, -1) for t in ts[m, 3]])
        busy, area, n, last = 0.0, 0.0, 0, t0
        for t, d in ev:
            if n > 0:
                busy += t - last
            area += n * (t - last)
            n += d
            last = t
        in_loop.append(busy / span)
        occ.append(area / span)
    print("  per CU: some block in its main loop %.0f %% of the span; mean blocks in the loop %.2f; blocks per CU %.1f" %
          (100 * np.mean(in_loop), np.mean(occ), len(a) / len(set(cu.tolist()))))
