#!/bin/bash
# usage: gpu_ab_env.sh VAR "v1 v2" "model1 model2" : interleaved A/B of an environment switch through bench.py
VAR=$1; VALS=$2; MODELS=$3
for m in $MODELS; do
  for rep in 1 2; do
    for v in $VALS; do
      r=$(env $VAR=$v timeout -k 10 240 python bench.py --model $m --steps 40 --warmup 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || exit 1
      echo "$m $VAR=$v $r"
    done
  done
done
