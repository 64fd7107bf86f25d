set -o pipefail
R=$PWD; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
# 1. bench lines
timeout -k 10 500 python bench.py > $O/bench_main.log 2>$O/bench_main.err || { echo BENCH FAIL; tail -3 $O/bench_main.err; exit 1; }
tail -1 $O/bench_main.log > $O/r03_bench_line.json
timeout -k 10 300 python bench.py --rollout 256 --no-strict-fp32 --no-cpu-baseline > $O/b256.log 2>/dev/null && tail -1 $O/b256.log > $O/r03_bench_line_ns2d_T256.json
timeout -k 10 300 python bench.py --preset sw_96x192x5 --no-strict-fp32 --no-cpu-baseline > $O/bsw.log 2>/dev/null && tail -1 $O/bsw.log > $O/r03_bench_line_sw_96x192x5.json
timeout -k 10 300 python bench.py --preset twophase_cond --batch 32 --rollout 128 --no-strict-fp32 --no-cpu-baseline > $O/btp.log 2>/dev/null && tail -1 $O/btp.log > $O/r03_bench_line_twophase_cond.json
echo bench done
cd /tmp && export TMPDIR=/tmp
# 2. kernel stats: serial (the per-kernel roofline pass) and overlapped
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/bench.py --serial --steps 8 --warmup 1 --no-cpu-baseline --no-strict-fp32 --no-check > $O/serial.log 2>&1 || { echo SERIAL FAIL; tail -3 $O/serial.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/overl -- python3 $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-strict-fp32 --no-check > $O/overl.log 2>&1 || { echo OVERL FAIL; exit 1; }
echo stats done
# 3. PMC passes (counters only, their own runs)
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --serial --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-strict-fp32 --no-check > $O/pmc_$n.log 2>&1 || { echo PMC FAIL $n; tail -3 $O/pmc_$n.log; exit 1; }
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --serial --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-strict-fp32 --no-check > $O/pmc_$c.log 2>&1 || { echo PMC FAIL $c; exit 1; }
done
echo pmc done
cd $R
python3 tools/pmc_summary.py $O/r03_pmc.json $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_SQ_WAIT_ANY $O/pmc_SQ_ACTIVE_INST_VALU $O/pmc_SQ_LDS_BANK_CONFLICT > $O/pmc_summary.txt 2>&1
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/make_traffic_json.py $F $W $O/r03_traffic.json > $O/traffic.txt 2>&1
find $O -name "*kernel_stats.csv" | head; 
# keep the merge small: drop raw traces
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O
