#!/bin/bash
# fused BN backward finalize on the models with more, smaller BatchNorm layers / a host-bound step
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; v=$2; shift 2; echo "$name fused=$v $(DTF_BN_FUSED_FINALIZE=$v timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 "$@" 2>gpurun_out/r3y.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'))")" | tee -a gpurun_out/r3y_matrix.txt; }
: > gpurun_out/r3y_matrix.txt
run inception_v3 0 --model inception_v3 && run inception_v3 1 --model inception_v3 && run inception_v3 0 --model inception_v3 && run inception_v3 1 --model inception_v3 && \
run resnet152 0 --model resnet50 --model-arg depth=152 && run resnet152 1 --model resnet50 --model-arg depth=152 && \
run resnet101 0 --model resnet50 --model-arg depth=101 && run resnet101 1 --model resnet50 --model-arg depth=101 || { tail gpurun_out/r3y.err; exit 1; }
