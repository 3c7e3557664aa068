#!/bin/bash
# K13 on / off through bench.py for the convolution models (same box, alternating)
set -o pipefail
out=gpurun_out/k13_models_ab.txt; : > $out
run() { name=$1; mode=$2; shift 2; echo "$name DTF_CONV_FWD=$mode $(DTF_CONV_FWD=$mode timeout -k 10 300 python -u bench.py "$@" --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['unit'], j['ms_per_step'], 'ms/step')")" | tee -a $out; }
for rep in 1 2; do
run inception_v3 0 --model inception_v3 && run inception_v3 auto --model inception_v3 || exit 1
done
run resnet101 0 --model resnet50 --model-arg depth=101 && run resnet101 auto --model resnet50 --model-arg depth=101 && \
run resnet152 0 --model resnet50 --model-arg depth=152 && run resnet152 auto --model resnet50 --model-arg depth=152
