# same-box timing of single 3x3 layers: shipped tiles (11) vs the producer / consumer forms (15, 16), and timing what-if
# builds of the latter (build/variants/pcknock<bits>: 1 no global loads, 2 no transform, 4 no MFMAs, 8 no patch gathers, 16 no weight loads)
set -o pipefail
R=$PWD; O=$R/gpurun_out/pc_time; rm -rf $O; mkdir -p $O
CASES="${CASES:-dec13 c64 c64_128 c32 c32_64 up32 lat lat_d2}"
cd /tmp && export TMPDIR=/tmp
run() {   # tag variant [lib]
  if [ -n "$3" ]; then export LNS_HIP_LIB=$3; else unset LNS_HIP_LIB; fi
  CONV_VARIANT=$2 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/$1 -- python3 $R/tools/conv_time.py $CASES > $O/$1.log 2>&1 || { echo FAIL $1; tail -5 $O/$1.log; exit 1; }
  echo "$1: $(python3 $R/tools/conv_time.py --parse $O/$1 $CASES)" | tee -a $O/summary.txt
}
run v11 11; run v15 15; run v16 16
for k in ${KNOCKS:-1 2 4 8 16}; do
  L=$R/build/variants/pcknock$k/pkg/liblns_hip.so
  if [ -f $L ]; then run v16_knock$k 16 $L; run v15_knock$k 15 $L; fi
done
find $O -name "*kernel_trace.csv" -size +1M -delete
