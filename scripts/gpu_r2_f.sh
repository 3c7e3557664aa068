#!/bin/bash
# round 2, call F: BatchNorm traversal-order A/B (Infinity Cache reuse between the two passes over a tensor)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "bn or batchnorm or resnet or inception" 2>&1 | tail -2 || exit 1
for rev in 0 2 10 15 5 8 1 4 3 12 6 9; do
  echo "DTF_BN_REV=$rev"
  DTF_BN_REV=$rev timeout -k 10 200 python -u bench.py --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
done 2>&1 | tee gpurun_out/bn_rev_ab.log
