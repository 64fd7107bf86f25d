#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/s11; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv_kernel or same_bits or domain or golden or every_layer or oct8" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
bash tools/ab_time.sh s11_layers ab/liblns_hip_noswz3.so main || exit 1
bash tools/ab_rollout.sh s11_rollout ab/liblns_hip_noswz3.so main || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lds -- python3 $R/bench.py --serial --steps 1 --warmup 0 --no-cpu-baseline --no-strict-fp32 --no-check --no-check-stable --no-rccl-world1 --no-roofline > $O/pmc_lds.log 2>&1 || { echo PMC FAIL; tail -3 $O/pmc_lds.log; exit 1; }
cd $R; python3 tools/pmc_summary.py $O/pmc_lds.json $O/pmc_lds 2>&1 | grep "conv3" | cut -c1-250
find $O -name "*counter_collection.csv" -size +1M -delete; find $O -name "*kernel_trace.csv" -size +1M -delete
