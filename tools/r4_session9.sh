#!/bin/bash
# experimental-build tests once (the -DLNS_EXPERIMENTAL library through LNS_HIP_LIB), then the full suite on the shipped build
set -o pipefail
R=$PWD; O=$R/gpurun_out/s9; rm -rf $O; mkdir -p $O
LNS_HIP_LIB=$R/ab/liblns_hip_exp.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conv_kernel or same_bits or quad_phase" > $O/exp_tests.log 2>&1
rc=$?; echo "experimental-build tests rc=$rc"; tail -3 $O/exp_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
rc=$?; echo "shipped-build tests rc=$rc"; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
