#!/bin/bash
# the five model configurations through bench.py on one GPU -> gpurun_out/all_models.txt
set -o pipefail
ENVV=DTF_NOOP=1
out=gpurun_out/all_models.txt; : > $out
run() { name=$1; shift; echo "$name $(env $ENVV timeout -k 10 300 python -u bench.py "$@" --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['unit'], j['ms_per_step'], 'ms/step host_enqueue', j['config'].get('host_enqueue_ms_per_step'))")" | tee -a $out; }
run resnet50 --model resnet50 && ENVV=DTF_FORCE_COMM=1 run resnet50_rccl1 --model resnet50 && \
run resnet101 --model resnet50 --model-arg depth=101 && run resnet152 --model resnet50 --model-arg depth=152 && \
run inception_v3 --model inception_v3 && run bert --model bert && run wide_deep --model wide_deep
