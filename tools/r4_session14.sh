#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/s14; rm -rf $O; mkdir -p $O
line() { tail -1 "$1" > "$2"; python3 -c "import json,sys; d=json.load(open('$2')); print('$3', round(d['value']), 'traj-steps/s', round(d['ms_per_step'],2), 'ms  frac', round(d['roofline']['frac'],3), 'check', d.get('check',{}).get('pass'), 'stable', d.get('check_stable',{}).get('pass'))"; }
timeout -k 10 900 python bench.py > $O/bench_main.log 2>$O/bench_main.err || { echo FAIL main; tail -3 $O/bench_main.err; exit 1; }
line $O/bench_main.log $O/r04_bench_line.json headline
timeout -k 10 400 python bench.py --rollout 256 --no-strict-fp32 --no-cpu-baseline --no-rccl-world1 > $O/b256.log 2>$O/b256.err || exit 1
line $O/b256.log $O/r04_bench_line_ns2d_T256.json T256
timeout -k 10 400 python bench.py --preset sw_96x192x5 --no-strict-fp32 --no-cpu-baseline --no-rccl-world1 > $O/bsw.log 2>$O/bsw.err || exit 1
line $O/bsw.log $O/r04_bench_line_sw_96x192x5.json sw
timeout -k 10 400 python bench.py --preset twophase_cond --batch 32 --rollout 128 --no-strict-fp32 --no-cpu-baseline --no-rccl-world1 > $O/btp.log 2>$O/btp.err || exit 1
line $O/btp.log $O/r04_bench_line_twophase_cond.json twophase_cond
