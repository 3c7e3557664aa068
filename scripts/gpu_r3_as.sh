#!/bin/bash
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "bn" 2>&1 | tail -2 || exit 1
run() { name=$1; shift; echo "$name $(timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host_probe', c['host_enqueue_ms_per_step'], 'host_loop', c['host_loop_ms_per_step'])")"; }
run resnet152 --model resnet50 --model-arg depth=152 && run resnet152 --model resnet50 --model-arg depth=152 && run resnet101 --model resnet50 --model-arg depth=101 && run resnet50 && run inception_v3 --model inception_v3
