#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ah.txt
for spec in "64 bert" "256 inception_v3" "256 resnet50"; do
  for ws in 1 0; do
    echo "== $spec DTF_WGRAD_STREAM=$ws" | tee -a gpurun_out/r3ah.txt
    DTF_WGRAD_STREAM=$ws timeout -k 10 400 python scripts/determinism_check.py $spec 2>&1 | grep -E "^iteration|^loss|Error|error" | cut -c1-420 | tee -a gpurun_out/r3ah.txt || exit 1
  done
done
