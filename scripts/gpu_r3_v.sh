#!/bin/bash
# deeper / wider models are close to host-bound in eager mode: graph replay vs eager per model
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; g=$2; shift 2; echo "$name graph=$g $(timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 --graph $g "$@" 2>gpurun_out/r3v.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), 'graph', c.get('hipgraph'), 'queues', c.get('gpu_max_hw_queues'), c.get('hipgraph_trial'))")" | tee -a gpurun_out/r3v_matrix.txt; }
: > gpurun_out/r3v_matrix.txt
run resnet152 0 --model resnet50 --model-arg depth=152 && run resnet152 1 --model resnet50 --model-arg depth=152 && \
run resnet101 0 --model resnet50 --model-arg depth=101 && run resnet101 1 --model resnet50 --model-arg depth=101 && \
run inception_v3 0 --model inception_v3 && run inception_v3 1 --model inception_v3 && \
run bert 0 --model bert && run bert 1 --model bert || { tail gpurun_out/r3v.err; exit 1; }
