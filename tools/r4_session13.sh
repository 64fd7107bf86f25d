#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-strict-fp32 --no-check --no-check-stable --no-rccl-world1"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --serial --steps 4 --warmup 1 $QUIET > $O/st.log 2>&1 || { echo FAIL; tail -3 $O/st.log; exit 1; }
cd $R
grep '^{"metric"' $O/st.log | tail -1 > $O/st.json
python3 tools/roofline_from_stats.py $(find $O/st -name "*kernel_stats.csv" | head -1) $O/st.json 6 | cut -c1-200
python3 -c "import json; d=json.load(open('$O/st.json')); print('overhead us', d['roofline']['event_overhead_us_subtracted_per_launch'])"
timeout -k 10 300 python bench.py $QUIET > $O/b.log 2>/dev/null && python3 -c "import json; d=json.loads(open('$O/b.log').read().strip().splitlines()[-1]); print('unprofiled: frac', d['roofline']['frac'], 'overhead us', d['roofline']['event_overhead_us_subtracted_per_launch'], 'nine-tap ms', [e['ms'] for e in d['roofline']['entries'] if e['form']=='f16x2 3x3 nine-tap'])"
find $O -name "*kernel_trace.csv" -size +1M -delete
