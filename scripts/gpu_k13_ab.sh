#!/bin/bash
# K13 (own implicit-GEMM convolution) end to end: tests, then ResNet-50 bench A/B in one box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "conv or gemm or linear" > gpurun_out/k13_tests.log 2>&1 || { tail -30 gpurun_out/k13_tests.log; exit 1; }
tail -2 gpurun_out/k13_tests.log
timeout -k 10 500 python -u -m pytest tests/test_models_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/k13_models.log 2>&1 || { tail -30 gpurun_out/k13_models.log; exit 1; }
tail -2 gpurun_out/k13_models.log
b() { DTF_CONV_FWD=$1 DTF_CONV_FWD_MIN_CH=$2 timeout -k 10 200 python bench.py --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('resnet50 conv_fwd=$1 min_ch=$2', r['value'], r['ms_per_step'])"; }
b 0 160 && b auto 160 && b auto 128 && b 0 160 && b auto 160 && b auto 128
