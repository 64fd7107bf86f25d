#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/s6; rm -rf $O; mkdir -p $O
SWEEP_GROUPS=1,2 SWEEP_STREAMS=2,3,4 timeout -k 10 400 python tools/sched_sweep.py ns2d_128 64 64 4 > $O/ns2d.jsonl 2> $O/ns2d.err
rc=$?; echo "ns2d rc=$rc"; cat $O/ns2d.jsonl
[ $rc -eq 0 ] && SWEEP_GROUPS=1,2,4 SWEEP_STREAMS=2,3,4 timeout -k 10 400 python tools/sched_sweep.py twophase_cond 32 128 4 > $O/tp.jsonl 2> $O/tp.err
rc=$?; echo "tp rc=$rc"; cat $O/tp.jsonl
[ $rc -eq 0 ] && SWEEP_GROUPS=1,2 SWEEP_STREAMS=2,3,4 timeout -k 10 400 python tools/sched_sweep.py sw_96x192x5 64 64 3 > $O/sw.jsonl 2> $O/sw.err
rc=$?; echo "sw rc=$rc"; cat $O/sw.jsonl
