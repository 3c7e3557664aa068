#!/bin/bash
# weight gradients on a side stream: A/B on ResNet-50, Inception-v3, BERT
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for m in resnet50 inception_v3; do
for v in 1 0 1 0; do
  DTF_WGRAD_STREAM=$v timeout -k 10 300 python bench.py --model $m --steps 30 --warmup 10 --graph 0 2> gpurun_out/r3h_bench_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m WGRAD_STREAM=$v', d['value'], d['ms_per_step'], d['config']['final_loss'])" || { tail gpurun_out/r3h_bench_$v.err; exit 1; }
done; done | tee gpurun_out/r3h_ab.txt
