#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ad.txt
for cfg in "$@"; do
  echo "== $cfg" | tee -a gpurun_out/r3ad.txt
  env $cfg timeout -k 10 300 python scripts/determinism_check.py 256 2>&1 | grep -E "^iteration|^loss|wgrad debug" | cut -c1-330 | tee -a gpurun_out/r3ad.txt || exit 1
done
