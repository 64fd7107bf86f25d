#!/usr/bin/env python3
"""Exports the two summaries kept under profiles/ from rocprofv3's rocpd (sqlite) output:

    python tools/rocpd_export.py kernel_stats <results.db> <out.csv>     # the --kernel-trace --stats table
    python tools/rocpd_export.py counters <results.db> <out.csv>         # per-dispatch counter values (--pmc pass)
"""
import csv
import math
import sqlite3
import sys


def kernel_stats(db, out):
    c = sqlite3.connect(db)
    rows = {}
    for name, dur in c.execute("select name, duration from kernels"):
        rows.setdefault(name, []).append(float(dur))
    tot = sum(sum(v) for v in rows.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            n, s = len(v), sum(v)
            mean = s / n
            sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
            w.writerow([name, n, int(s), round(mean, 3), round(100.0 * s / tot, 4), int(min(v)), int(max(v)), round(sd, 3)])


def counters(db, out):
    c = sqlite3.connect(db)
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for r in c.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value "
                           "from counters_collection order by dispatch_id"):
            w.writerow(list(r))


if __name__ == "__main__":
    {"kernel_stats": kernel_stats, "counters": counters}[sys.argv[1]](sys.argv[2], sys.argv[3])
