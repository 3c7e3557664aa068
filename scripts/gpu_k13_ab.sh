#!/bin/bash
# K13 (own implicit-GEMM convolution) end to end: tests, then ResNet-50 / Inception-v3 bench A/B in one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "conv_fwd_implicit or conv_module_on_k13 or filter_flip or conv_data_gradient or conv_wrw" > gpurun_out/k13_tests.log 2>&1 || { tail -30 gpurun_out/k13_tests.log; exit 1; }
tail -2 gpurun_out/k13_tests.log
timeout -k 10 500 python -u -m pytest tests/test_models_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/k13_models.log 2>&1 || { tail -30 gpurun_out/k13_models.log; exit 1; }
tail -2 gpurun_out/k13_models.log
for mode in 0 auto 0 auto; do
  for st in 1; do
    DTF_CONV_FWD=$mode DTF_CONV_FWD_STATS=$st timeout -k 10 200 python bench.py --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('resnet50 conv_fwd=$mode stats=$st', r['value'], r['ms_per_step'])" || exit 1
  done
done
DTF_CONV_FWD=auto DTF_CONV_FWD_STATS=0 timeout -k 10 200 python bench.py --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('resnet50 conv_fwd=auto stats=0', r['value'], r['ms_per_step'])" || exit 1
