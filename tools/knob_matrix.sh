#!/bin/bash
# The alternative paths behind the library's A/B knobs still pass the model-level parity tests (run on the GPU box):
#   bash tools/knob_matrix.sh     -> gpurun_out/knob_matrix.txt
R=$PWD; O=$R/gpurun_out/knob_matrix.txt; : > $O
K="every_layer or rollout_matches_reference_golden or batch_shard or overlapped_rollout_equals"
for kv in "LNS_CONV3_SPLIT=bf16x3" "LNS_GN_NO_FOLD=1" "LNS_GN_NO_RAGGED=1" "LNS_GN_NO_FUSE=1" "LNS_NO_UP2_PHASES=1" "LNS_UP2_RESIDENT=1" "LNS_NO_ARITH_MAPS=1" \
          "LNS_CONVF32_BELOW=100000" "LNS_CONVF32_BELOW=0" "LNS_NO_GELU_PROLOGUE=1" "LNS_NO_FUSE_1X1=1" "LNS_CONV_FP32_MFMA=1"; do
  res=$(env $kv timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" 2>&1 | tail -1)
  echo "$kv: $res" | tee -a $O
done
