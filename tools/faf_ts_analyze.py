#!/usr/bin/env python3
"""Phase timestamps of the fused FABlock kernel (fa_fused2_kernel in a `make DIAGFLAGS=-DFAF_TS` build, LNS_TS_FILE): where a
block's time goes.  Run with fa_fused_gpb = 1 (LNS_FA_FUSED_GPB=1): one plane group per block, 12 stamps.

    python tools/faf_ts_analyze.py <ts file>      (uses the last launch in the file)
slots: 0 entry, 1 Kx / Ky images written, 2 in_proj of band 0 done, then per band (4x): before the barrier, after it; 11 stores issued
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
t = a[:, :12].astype(np.float64) * 0.01        # us
t0 = t[:, 0].min()
dur = t[:, 11] - t[:, 0]
print("blocks %d   launch span %.1f us   block lifetime: median %.1f us  p10 %.1f  p90 %.1f" % (len(a), t[:, 11].max() - t0, np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90)))
print("prologue (Kx, Ky -> bounds -> images) %.2f us;  in_proj of band 0 %.2f us" % (np.median(t[:, 1] - t[:, 0]), np.median(t[:, 2] - t[:, 1])))
for jb in range(4):
    base = 3 + 2 * jb
    end = t[:, base + 2] if jb < 3 else None
    msg = "band %d: wait at the barrier %.2f" % (jb, np.median(t[:, base + 1] - t[:, base]))
    if end is not None:
        msg += "   sandwich + in_proj of the next band %.2f" % np.median(end - t[:, base + 1])
    print(msg)
print("sandwich of band 3 + epilogue (norm + stores issued) %.2f us" % np.median(t[:, 11] - t[:, 10]))
start = t[:, 0] - t0
print("block start times by decile:", np.round(np.percentile(start, [10, 30, 50, 70, 90]), 1))
