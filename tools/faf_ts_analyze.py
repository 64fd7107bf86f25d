#!/usr/bin/env python3
"""Phase timestamps of fa_fused_kernel (a -DFAF_TS build, LNS_TS_FILE): where a block's time goes.

    python tools/faf_ts_analyze.py <ts file>      (uses the last launch in the file)
slots: 0 entry, 1 Kx/Ky requested, 2 band-0 in_proj + Kx/Ky images done, then per band (4x): before the barrier, after it, B done;
15 stores issued; 23 XCC id
"""
import sys
import numpy as np

rows, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        if cur:
            rows = cur
        cur = []
        continue
    cur.append([int(v) for v in line.split()[1:]])
if cur:
    rows = cur
a = np.array(rows, dtype=np.int64)
t = a[:, :16].astype(np.float64) * 0.01        # us
t0 = t[:, 0].min()
dur = t[:, 15] - t[:, 0]
print("blocks %d   launch span %.1f us   block lifetime: median %.1f us  p10 %.1f  p90 %.1f" % (len(a), t[:, 15].max() - t0, np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90)))
print("entry -> Kx/Ky requested, scales %.2f us;  in_proj of band 0 + Kx/Ky bounds and images %.2f us" % (np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1])))
for jb in range(4):
    base = 3 + 3 * jb
    nxt = t[:, base + 3] if jb < 3 else None
    msg = "band %d: wait at the barrier %.2f   B (sandwich) %.2f" % (jb, np.median(t[:, base + 1] - t[:, base]), np.median(t[:, base + 2] - t[:, base + 1]))
    if nxt is not None:
        msg += "   barrier + A (in_proj of the next band) %.2f" % np.median(nxt - t[:, base + 2])
    print(msg)
print("last barrier + epilogue (norm + stores issued) %.2f us" % np.median(t[:, 15] - t[:, 14]))
start = t[:, 0] - t0
print("block start times by decile:", np.round(np.percentile(start, [10, 30, 50, 70, 90]), 1))
