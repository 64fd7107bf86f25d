#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/s5; rm -rf $O; mkdir -p $O
SWEEP_GROUPS=1 SWEEP_STREAMS=3 SWEEP_SERIAL=1 SWEEP_FA_CHUNK_MB=0,256,128,64,32 timeout -k 10 300 python tools/sched_sweep.py ns2d_128 64 64 3 > $O/serial.jsonl 2> $O/serial.err
rc=$?; echo "serial rc=$rc"; cat $O/serial.jsonl; tail -3 $O/serial.err | cut -c1-200
[ $rc -eq 0 ] && SWEEP_GROUPS=1 SWEEP_STREAMS=3 SWEEP_FA_CHUNK_MB=0,256,128,64,32,16 timeout -k 10 300 python tools/sched_sweep.py ns2d_128 64 64 3 > $O/overlap.jsonl 2> $O/overlap.err
rc=$?; echo "overlap rc=$rc"; cat $O/overlap.jsonl; tail -3 $O/overlap.err | cut -c1-200
