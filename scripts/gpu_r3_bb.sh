#!/bin/bash
# K21 after the lane-per-pixel patch kernel and the grouped-rows product: tests, Inception-v3 A/B against MIOpen's kernel
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3bb.txt
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "image_input or foreign_layout" 2>&1 | tail -3 || exit 1
timeout -k 10 400 python -u scripts/determinism_check.py 256 inception_v3 2>gpurun_out/r3bb.err | tail -1 | tee -a gpurun_out/r3bb.txt || exit 1
for v in 1 0 1 0; do echo "bench inception_v3 K21=$v $(DTF_CONV_IMAGE_WRW=$v timeout -k 10 400 python -u bench.py --model inception_v3 --steps 30 --warmup 8 2>>gpurun_out/r3bb.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")" | tee -a gpurun_out/r3bb.txt; done
