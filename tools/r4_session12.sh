#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/s12; rm -rf $O; mkdir -p $O
export LNS_HIP_LIB=$R/ab/liblns_hip_ts3.so LNS_TS_MAX=3
for L in "decoder.model.10.conv1" "decoder.model.3.conv1" "propagator.net.1.conv.3" "decoder.model.15" "decoder.model.11.in_proj" "decoder.model.11.to_out"; do
  # each layer runs once per step: skip 20 rollouts x 64 steps, then stamp three launches
  LNS_TS_FILE=$O/ts_$L.txt LNS_TS_LAYER=$L LNS_TS_SKIP=1290 timeout -k 10 200 python tools/clock_probe.py ns2d_128 64 22 > $O/probe_$L.log 2>&1 || { echo "probe $L failed"; tail -3 $O/probe_$L.log; exit 1; }
  python3 tools/clock_analyze.py $O/ts_$L.txt | tee -a $O/clock.txt
done
rm -f $O/ts_*.txt
