#!/bin/bash
# DTF_DETERMINISTIC as the default?  A/B per model
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ag.txt
run() { name=$1; v=$2; shift 2; echo "$name det=$v $(DTF_DETERMINISTIC=$v timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 "$@" 2>gpurun_out/r3ag.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")" | tee -a gpurun_out/r3ag.txt; }
run resnet50 1 && run resnet50 0 && run resnet50 1 && run resnet50 0 && \
run inception_v3 1 --model inception_v3 && run inception_v3 0 --model inception_v3 && \
run resnet152 1 --model resnet50 --model-arg depth=152 && run resnet152 0 --model resnet50 --model-arg depth=152 && \
run bert 1 --model bert && run bert 0 --model bert || { tail gpurun_out/r3ag.err; exit 1; }
