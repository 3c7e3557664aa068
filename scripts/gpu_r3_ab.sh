#!/bin/bash
# run-to-run differences of the ResNet-50 gradients at batch 256: side stream on / off, library weight gradients on / off
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "DTF_NOOP=1" "DTF_WGRAD_STREAM=0" "DTF_WGRAD_STREAM=0 DTF_CONV1X1_WRW_LIB=0 DTF_GEMM_TUNING=0"; do
  echo "== $cfg" | tee -a gpurun_out/r3ab.txt
  env $cfg timeout -k 10 300 python scripts/determinism_check.py 256 2>&1 | grep -E "^iteration|^loss" | cut -c1-700 | tee -a gpurun_out/r3ab.txt || exit 1
done
