#!/bin/bash
# A/B of the K6T conv weight gradient inside the two CNN families
set -e
mkdir -p gpurun_out
for m in resnet50 inception_v3; do
  for v in 0 1 0 1; do
    echo "== $m DTF_CONV_WRW=$v"
    DTF_CONV_WRW=$v timeout -k 10 240 python bench.py --model $m --steps 40 --warmup 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab_convwrw.log
  done
done
