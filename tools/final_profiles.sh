#!/bin/bash
# The round's committed measurement set, one box, one build (run on the GPU box from the repo root):
#   tools/final_profiles.sh r04            -> gpurun_out/final/r04_*  (copy into profiles/ afterwards)
# bench lines of the four single-GPU configurations, rocprofv3 kernel stats of the single-stream pass of each (what
# `roofline.frac` must reproduce from, tools/roofline_from_stats.py) and of the overlapped headline run, PMC passes (each in its
# own run: counters only with --kernel-trace), HBM traffic.  Stops at the first failing GPU step.
set -o pipefail
TAG=${1:-r04}
R=$PWD; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
fail() { echo "FAILED: $1"; tail -5 "$2" 2>/dev/null; exit 1; }
line() { tail -1 "$1" > "$2"; python3 -c "import json,sys; d=json.load(open('$2')); print('$3', round(d['value']), 'traj-steps/s', round(d['ms_per_step'],2), 'ms  frac', round(d['roofline']['frac'],3), 'check', d.get('check',{}).get('pass'), 'stable', d.get('check_stable',{}).get('pass'))"; }

cd /tmp && export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-strict-fp32 --no-check --no-check-stable --no-rccl-world1"
# 2. kernel stats of the single-stream pass, every configuration (steps 4 + warmup 1 + the roofline pass = 6 rollouts)
stats() {   # name, bench args...
  local n=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$n -- python3 $R/bench.py --serial --steps 4 --warmup 1 $QUIET "$@" > $O/st_$n.log 2>&1 || fail "stats $n" $O/st_$n.log
  local f=$(find $O/st_$n -name "*kernel_stats.csv" | head -1); cp $f $O/${TAG}_serial_kernel_stats$n.csv
  grep '^{"metric"' $O/st_$n.log | tail -1 > $O/st_$n.json
}
stats ""
stats _ns2d_T256 --rollout 256
stats _sw_96x192x5 --preset sw_96x192x5
stats _twophase_cond --preset twophase_cond --batch 32 --rollout 128
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/overl -- python3 $R/bench.py --steps 8 --warmup 1 $QUIET --no-roofline > $O/overl.log 2>&1 || fail "stats overlapped" $O/overl.log
cp $(find $O/overl -name "*kernel_stats.csv" | head -1) $O/${TAG}_overlapped_kernel_stats.csv
echo "stats done"
# 3. PMC passes (counters only, their own runs)
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --serial --steps 1 --warmup 0 $QUIET --no-roofline > $O/pmc_$n.log 2>&1 || fail "pmc $n" $O/pmc_$n.log
done
echo "pmc done"
cd $R
python3 tools/pmc_summary.py $O/${TAG}_pmc.json $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_SQ_WAIT_ANY $O/pmc_SQ_ACTIVE_INST_VALU $O/pmc_SQ_LDS_BANK_CONFLICT > $O/${TAG}_pmc_summary.txt 2>&1
F=$(ls -t $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1); W=$(ls -t $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1)
python3 tools/make_traffic_json.py $F $W $O/${TAG}_traffic.json > $O/traffic.txt 2>&1
# the bench lines quote the counter passes of THIS box and build: put them where bench.py looks (profiles/ of this copy)
cp $O/${TAG}_pmc.json $O/${TAG}_traffic.json $R/profiles/
# 3b. bench lines
timeout -k 10 900 python bench.py > $O/bench_main.log 2>$O/bench_main.err || fail "bench main" $O/bench_main.err
line $O/bench_main.log $O/${TAG}_bench_line.json headline
timeout -k 10 400 python bench.py --rollout 256 --no-strict-fp32 --no-cpu-baseline --no-rccl-world1 > $O/b256.log 2>$O/b256.err || fail "bench T256" $O/b256.err
line $O/b256.log $O/${TAG}_bench_line_ns2d_T256.json T256
timeout -k 10 400 python bench.py --preset sw_96x192x5 --no-strict-fp32 --no-cpu-baseline --no-rccl-world1 > $O/bsw.log 2>$O/bsw.err || fail "bench sw" $O/bsw.err
line $O/bsw.log $O/${TAG}_bench_line_sw_96x192x5.json sw
timeout -k 10 400 python bench.py --preset twophase_cond --batch 32 --rollout 128 --no-strict-fp32 --no-cpu-baseline --no-rccl-world1 > $O/btp.log 2>$O/btp.err || fail "bench twophase" $O/btp.err
line $O/btp.log $O/${TAG}_bench_line_twophase_cond.json twophase_cond
echo "bench done"

# 4. the roofline block recomputed from the rocprofv3 stats, every configuration
for n in "" _ns2d_T256 _sw_96x192x5 _twophase_cond; do
  echo "== ${TAG}_serial_kernel_stats$n.csv vs the bench line of the same run" >> $O/${TAG}_roofline_from_stats.txt
  python3 tools/roofline_from_stats.py $O/${TAG}_serial_kernel_stats$n.csv $O/st_$n.json 6 >> $O/${TAG}_roofline_from_stats.txt 2>&1
done
cat $O/${TAG}_roofline_from_stats.txt | grep -i "nine-tap \|rollouts\|==" | cut -c1-220
# keep the merge small: drop raw traces
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*counter_collection.csv" -size +1M -delete
du -sh $O
