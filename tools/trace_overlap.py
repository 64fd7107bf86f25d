#!/usr/bin/env python3
"""How well does the multi-stream rollout fill the GPU?  From a rocprofv3 kernel trace of `bench.py`:
union of the kernel intervals (some kernel running), sum of durations (serial kernel time), idle gaps.
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strict-fp32 --no-check
    python tools/trace_overlap.py out"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# drop the warm-up: keep the last 60 % of the time span
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + 0.4 * (t1 - t0)
rows = [r for r in rows if r[0] >= cut]
span = max(r[1] for r in rows) - rows[0][0]
tot = sum(e - s for s, e, _ in rows)
union, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
for s, e, _ in rows[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("span %.2f ms, some kernel running %.1f %%, sum of kernel durations / span = %.2f (mean kernels in flight)" %
      (span / 1e6, 100.0 * union / span, tot / span))
print("idle gaps: %d, total %.2f ms, largest %.1f us" % (len(gaps), sum(gaps) / 1e6, max(gaps) / 1e3 if gaps else 0))
# time with exactly one kernel running vs more
ev = sorted([(s, 1) for s, e, _ in rows] + [(e, -1) for s, e, _ in rows])
hist, n, last = {}, 0, ev[0][0]
for t, d in ev:
    hist[n] = hist.get(n, 0) + (t - last)
    n += d
    last = t
for k in sorted(hist):
    print("  %d kernels in flight: %5.1f %% of the span" % (k, 100.0 * hist[k] / span))
# which kernels run ALONE (time with exactly one kernel in flight, by kernel) -- small ones there are idle CUs
import collections, re
ev = sorted([(s, 1, i) for i, (s, e, _) in enumerate(rows)] + [(e, -1, i) for i, (s, e, _) in enumerate(rows)])
alone = collections.Counter()
live, last = set(), ev[0][0]
for t, d, i in ev:
    if len(live) == 1:
        alone[re.sub(r"\(.*", "", rows[next(iter(live))][2])[:70]] += t - last
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
    last = t
tot_alone = sum(alone.values())
print("alone-time by kernel (%.1f %% of the span):" % (100.0 * tot_alone / span))
for k, v in alone.most_common(14):
    print("  %5.1f %%  %s" % (100.0 * v / span, k))
