#!/bin/bash
# BatchNorm statistics in the 1x1 GEMM epilogue (K6 / K13 pipeline) re-measured with this round's kernels: A/B per threshold
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3at.txt
run() { name=$1; st=$2; mc=$3; shift 3; echo "$name stats=$st min_cout=$mc $(DTF_CONV1X1_STATS=$st DTF_CONV1X1_STATS_MIN_COUT=$mc timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 "$@" 2>gpurun_out/r3at.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")" | tee -a gpurun_out/r3at.txt; }
run resnet50 0 256 && run resnet50 1 256 && run resnet50 1 512 && run resnet50 1 1024 && run resnet50 1 2048 && run resnet50 0 256 || { tail gpurun_out/r3at.err; exit 1; }
