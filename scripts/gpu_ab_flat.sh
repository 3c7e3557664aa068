#!/bin/bash
# A/B of the flat K7 kernels through the whole ResNet-50 / Inception-v3 step.  usage: gpu_ab_flat.sh VAR "values" "models" reps
set -o pipefail
var=${1:-DTF_BN_FLAT_STATS}; vals=${2:-"0 1 2"}; models=${3:-"resnet50 inception_v3"}; reps=${4:-2}
out=gpurun_out/ab_${var}.txt; : > $out
for rep in $(seq $reps); do
for mode in $vals; do
  for model in $models; do
    v=$(env $var=$mode timeout -k 10 200 python -u bench.py --model $model --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])") || exit 1
    echo "$var=$mode $model $v" | tee -a $out
  done
done
done
