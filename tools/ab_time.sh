#!/bin/bash
# Same-box A/B of single conv layers between library builds (rocprofv3 kernel trace, min over launches):
#   tools/ab_time.sh <out name> <lib A (path or "main")> <lib B> ...       run on the GPU box from the repo root
set -o pipefail
R=$PWD; name=$1; shift
O=$R/gpurun_out/$name; rm -rf $O; mkdir -p $O
C11="c64 c64_128 c32 lat f64 k1_lat k1_lat_up k1_lat_dn k1_toout k1_128 k1_toout_gelu_res"
C17="dec13 up32"
cd /tmp && export TMPDIR=/tmp
i=0
for lib in "$@"; do
  i=$((i+1))
  if [ "$lib" = main ]; then unset LNS_HIP_LIB; else export LNS_HIP_LIB=$R/$lib; fi
  CONV_VARIANT=11 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/l${i}v11 -- python3 $R/tools/conv_time.py $C11 > $O/l${i}v11.log 2>&1 || { echo FAIL $lib; tail -5 $O/l${i}v11.log; exit 1; }
  CONV_VARIANT=17 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/l${i}v17 -- python3 $R/tools/conv_time.py $C17 > $O/l${i}v17.log 2>&1 || { echo FAIL $lib; tail -5 $O/l${i}v17.log; exit 1; }
  echo "$lib: $(python3 $R/tools/conv_time.py --parse $O/l${i}v11 $C11) $(python3 $R/tools/conv_time.py --parse $O/l${i}v17 $C17)" | tee -a $O/summary.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
