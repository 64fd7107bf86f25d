#!/usr/bin/env python3
"""Per-queue (HIP stream) launch statistics of a rocprofv3 kernel trace: launches, busy time, span, and the distribution of
the gaps between consecutive kernels of the same queue.   python tools/queue_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys, collections
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
q = collections.defaultdict(list)
for r in rows:
    q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for qid, ks in sorted(q.items(), key=lambda kv: -len(kv[1])):
    ks.sort()
    busy = sum(e - s for s, e, _ in ks)
    gaps = sorted(max(0, ks[i][0] - ks[i - 1][1]) for i in range(1, len(ks)))
    n = len(gaps)
    if n < 10:
        continue
    small = [g for g in gaps if g < 20000]
    print("queue %s: %d launches, busy %.2f ms, span %.2f ms, mean kernel %.1f us; gaps: median %.2f us, p90 %.2f us, mean of gaps < 20 us %.2f us (%d of %d), sum of all gaps %.2f ms"
          % (qid, len(ks), busy / 1e6, (ks[-1][1] - ks[0][0]) / 1e6, busy / len(ks) / 1e3, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3,
             sum(small) / max(1, len(small)) / 1e3, len(small), n, sum(gaps) / 1e6))
