#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ac.txt
for cfg in "DTF_WGRAD_DEBUG=1" "DTF_WGRAD_DEBUG=2" "DTF_CONV1X1_TN=0" "DTF_KEEP_HW_QUEUES=1"; do
  echo "== $cfg" | tee -a gpurun_out/r3ac.txt
  env $cfg timeout -k 10 300 python scripts/determinism_check.py 256 2>&1 | grep -E "^iteration|^loss" | cut -c1-420 | tee -a gpurun_out/r3ac.txt || exit 1
done
