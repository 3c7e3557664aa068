#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "image_input or conv_wrw" 2>&1 | tail -4 || exit 1
echo "== inception determinism"; timeout -k 10 400 python scripts/determinism_check.py 256 inception_v3 2>&1 | grep -E "^iteration|^loss" | cut -c1-300
scripts/gpu_r3_ao.sh
