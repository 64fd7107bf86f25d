#!/bin/bash
# round 4, GPU session 3: FABlock chunking -- bitwise test, then sweeps (serial and overlapped)
set -o pipefail
R=$PWD; O=$R/gpurun_out/s3; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped_rollout_equals_single_stream or conv_kernel or oct8 or module_api" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
SWEEP_GROUPS=1 SWEEP_STREAMS=3 SWEEP_SERIAL=1 SWEEP_FA_CHUNK_MB=0,256,128,64,32 timeout -k 10 400 python tools/sched_sweep.py ns2d_128 64 64 4 2>&1 | tee $O/sweep_serial.jsonl | tail -8
SWEEP_GROUPS=1 SWEEP_STREAMS=3 SWEEP_FA_CHUNK_MB=0,256,128,64,32,16 timeout -k 10 400 python tools/sched_sweep.py ns2d_128 64 64 4 2>&1 | tee $O/sweep_overlap.jsonl | tail -8
SWEEP_GROUPS=1,2 SWEEP_STREAMS=2 SWEEP_FA_CHUNK_MB=0,128,64 timeout -k 10 400 python tools/sched_sweep.py ns2d_128 64 64 4 2>&1 | tee $O/sweep_overlap2.jsonl | tail -8
