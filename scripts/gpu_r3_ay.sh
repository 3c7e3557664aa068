#!/bin/bash
# K21 (image-input weight gradient = patch matrix + K6T) as the default route: numerics, Inception-v3 reproducibility and cost
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ay.txt
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "image_input or strict_mode or foreign_layout" 2>&1 | tail -15 || exit 1
echo "== inception_v3 batch 256 default mode" | tee -a gpurun_out/r3ay.txt
timeout -k 10 400 python -u scripts/determinism_check.py 256 inception_v3 2>gpurun_out/r3ay.err | tail -2 | cut -c1-300 | tee -a gpurun_out/r3ay.txt || { tail gpurun_out/r3ay.err; exit 1; }
for v in 1 0 1 0; do echo "bench inception_v3 K21=$v $(DTF_CONV_IMAGE_WRW=$v timeout -k 10 400 python -u bench.py --model inception_v3 --steps 30 --warmup 8 2>>gpurun_out/r3ay.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")" | tee -a gpurun_out/r3ay.txt; done
