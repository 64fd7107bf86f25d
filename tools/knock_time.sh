#!/bin/bash
# Timing what-if builds of the f16x2 3x3 kernel on single layers (results are garbage by construction):
#   tools/build_variant.sh knock1 -DLNS_KNOCK=1   (no staging inside the K loop)
#   tools/build_variant.sh knock3 -DLNS_KNOCK=3   (... and no fragment reads: MFMAs + barrier only)
#   bash tools/knock_time.sh        on the GPU box -> gpurun_out/knock/summary.txt
set -o pipefail
R=$PWD; O=$R/gpurun_out/knock; rm -rf $O; mkdir -p $O
C="lat lat_tp c32 c64_128"
cd /tmp && export TMPDIR=/tmp
n=0
for lib in main knock1 knock3 main knock1 knock3; do
  n=$((n+1))
  if [ $lib = main ]; then unset LNS_HIP_LIB; else export LNS_HIP_LIB=$R/build/variants/$lib/pkg/liblns_hip.so; fi
  CONV_VARIANT=11 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/$lib.$n -- python3 $R/tools/conv_time.py $C > $O/$lib.$n.log 2>&1 || { echo FAIL $lib; exit 1; }
  echo "$lib: $(python3 $R/tools/conv_time.py --parse $O/$lib.$n $C)" | tee -a $O/summary.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
