set -o pipefail
R=$PWD; O=$R/gpurun_out/up2_time; rm -rf $O; mkdir -p $O
CASES="dec13 up32"
cd /tmp && export TMPDIR=/tmp
n=0
for v in 11 17 18 17 18; do
  n=$((n+1))
  if [ $v = 18 ]; then CASES="dec13"; else CASES="dec13 up32"; fi      # (18 = resident-patch form: Cin_pad <= 64 only)
  CONV_VARIANT=$v timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/v$v.$n -- python3 $R/tools/conv_time.py $CASES > $O/v$v.$n.log 2>&1 || { echo FAIL v$v; tail -5 $O/v$v.$n.log; exit 1; }
  echo "variant $v: $(python3 $R/tools/conv_time.py --parse $O/v$v.$n $CASES)" | tee -a $O/summary.txt
done
find $O -name "*kernel_trace.csv" -size +1M -delete
