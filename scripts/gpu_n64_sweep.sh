#!/bin/bash
# K18 micro-benchmark over its tuning switches -> gpurun_out/n64_sweep2.txt
set -o pipefail
out=gpurun_out/n64_sweep2.txt; : > $out
for u in 1 2; do for bpc in 1 2; do
  echo "DTF_N64_NT=0 DTF_N64_U=$u DTF_N64_BLOCKS_PER_CU=$bpc" >> $out
  DTF_N64_NT=0 DTF_N64_U=$u DTF_N64_BLOCKS_PER_CU=$bpc timeout -k 10 100 python scripts/conv1x1_n64_bench.py 2>/dev/null | grep K= | cut -c1-48 >> $out || exit 1
done; done
cat $out
