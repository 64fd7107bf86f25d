#!/bin/bash
# quad-phase upsampling conv: bitwise tests, single-layer timing vs the per-phase form, rollout A/B
set -o pipefail
R=$PWD; O=$R/gpurun_out/s7; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "quad_phase or (conv_kernel and not oct) or oct8 or golden or overlapped" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log; [ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
for v in 17 20; do
  CONV_VARIANT=$v timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/ct_$v -- python3 $R/tools/conv_time.py dec13 up64 > $O/ct_$v.log 2>&1 || { echo "conv_time $v FAILED"; tail -3 $O/ct_$v.log; exit 1; }
  echo "variant $v: $(python3 $R/tools/conv_time.py --parse $O/ct_$v dec13 up64 2>&1 | tail -1)" | tee -a $O/conv_time_up2q.txt
done
cd $R
for rep in 1 2; do
  for arm in quad perphase; do
    if [ $arm = perphase ]; then export LNS_NO_UP2_QUAD=1; else unset LNS_NO_UP2_QUAD; fi
    timeout -k 10 300 python bench.py --no-strict-fp32 --no-cpu-baseline --no-rccl-world1 --no-check-stable > $O/b_${arm}_$rep.log 2>$O/b_${arm}_$rep.err || { echo "bench $arm FAILED"; tail -3 $O/b_${arm}_$rep.err; exit 1; }
    python3 - $arm $O/b_${arm}_$rep.log <<'PY' | tee -a $O/summary.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
e = {x["form"]: x for x in d["roofline"]["entries"]}
print("%-9s %8.0f traj-steps/s %7.2f ms check %s | " % (sys.argv[1], d["value"], d["ms_per_step"], d["check"]["pass"]) +
      "  ".join("%s: %.1f ms (%.0f us)" % (k[10:], v["ms"], v["avg_launch_us"]) for k, v in e.items() if "four-tap" in k))
PY
  done
done
find $O -name "*kernel_trace.csv" -size +1M -delete
