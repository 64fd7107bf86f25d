#!/bin/bash
# round 4, GPU session 1: full GPU suite, OCT8 layout A/B on single layers, bench with every sub-record
set -o pipefail
R=$PWD; O=$R/gpurun_out/s1; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc; tail -5 $O/tests.log
cd /tmp && export TMPDIR=/tmp
CASES="c64 c64_128 c32 c32_64 lat lat_d2 lat_tp"
for lay in 0 256 512 768; do
  CONV_LAYOUT=$lay timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/ct_$lay -- python3 $R/tools/conv_time.py $CASES > $O/ct_$lay.log 2>&1 || { echo "conv_time $lay FAILED"; tail -3 $O/ct_$lay.log; }
  echo "layout $lay: $(python3 $R/tools/conv_time.py --parse $O/ct_$lay $CASES 2>&1 | tail -1)" | tee -a $O/conv_time_oct8.txt
done
cd $R
timeout -k 10 900 python bench.py > $O/bench.log 2>$O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.err; tail -1 $O/bench.log > $O/bench_line.json
python3 -c "
import json;d=json.load(open('$O/bench_line.json'))
print('value',d['value'],'ms',d['ms_per_step'],'check',d.get('check',{}).get('pass'))
r=d.get('roofline',{});print('frac',r.get('frac'),'exec',r.get('executed_frac'),'pmc',r.get('mfma_busy_pmc'))
for e in r.get('entries',[]):print(e)
print(json.dumps(r.get('whole_path'),indent=1))
print('rccl',json.dumps(d.get('rccl_world1'))[:900])
print('strict',d.get('strict_fp32',{}).get('value'))
"
find $O -name "*kernel_trace.csv" -size +1M -delete
