#!/bin/bash
set -o pipefail
out=gpurun_out/conv1x1_256.txt; : > $out
d=$(mktemp -d /tmp/thr.XXXX); cp profiles/r2_conv1x1_all_shapes_resnet50.csv $d/t0.csv
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$d/t.csv PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_TUNING=0 DTF_CONV1X1_MM_TUNED=0
run() { echo "$* $(env "$@" timeout -k 10 200 python -u bench.py --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")" | tee -a $out; }
for rep in 1 2; do
run DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_DGRAD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=256
run DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_DGRAD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=256 DTF_CONV1X1_MM_FWD_SKIP_MAX_COUT=128
run DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_DGRAD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=256 DTF_CONV1X1_MM_FWD_SKIP_MAX_COUT=512
run DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_DGRAD_MIN_CIN=64
done
