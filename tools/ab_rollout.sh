#!/bin/bash
# Same-box A/B of the whole rollout between library builds: alternating `python bench.py` runs (headline workload, no side
# records), then one rocprofv3 --stats pass of the single-stream rollout per library.
#   tools/ab_rollout.sh <out name> <lib A (path or "main")> <lib B> ...      run on the GPU box from the repo root
set -o pipefail
R=$PWD; name=$1; shift
O=$R/gpurun_out/$name; rm -rf $O; mkdir -p $O
FLAGS="--no-strict-fp32 --no-cpu-baseline --no-rccl-world1 --no-check-stable ${AB_FLAGS:-}"
for rep in 1 2; do
  i=0
  for lib in "$@"; do
    i=$((i+1))
    if [ "$lib" = main ]; then unset LNS_HIP_LIB; else export LNS_HIP_LIB=$R/$lib; fi
    timeout -k 10 300 python bench.py $FLAGS > $O/b${i}_$rep.log 2>$O/b${i}_$rep.err || { echo FAIL $lib; tail -5 $O/b${i}_$rep.err; exit 1; }
    python3 - "$lib" $O/b${i}_$rep.log <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
kc = d.get("kernel_classes", {})
print("%-28s %8.0f traj-steps/s  %7.2f ms  check %s  | %s" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("check", {}).get("pass"),
      "  ".join("%s %.1f" % (k.replace("_mfma", ""), v["ms"]) for k, v in list(kc.items())[:6])))
PY
  done
done
cd /tmp && export TMPDIR=/tmp
i=0
for lib in "$@"; do
  i=$((i+1))
  if [ "$lib" = main ]; then unset LNS_HIP_LIB; else export LNS_HIP_LIB=$R/$lib; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$i -- python3 $R/bench.py --serial --steps 4 --warmup 1 --no-check --no-roofline $FLAGS > $O/st$i.log 2>&1 || { echo STATS FAIL $lib; tail -3 $O/st$i.log; }
  f=$(find $O/st$i -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/stats_$i.csv && echo "== $lib" >> $O/summary.txt && head -14 $O/stats_$i.csv | cut -d, -f1-4 >> $O/summary.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
cat $O/summary.txt
