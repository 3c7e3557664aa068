#!/bin/bash
# Which 1x1 convolution products should go through the GEMM library now that its solutions are chosen per shape?
# Tune all shapes once (forward through the library from 64 channels), then measure threshold / skip-list variants.
set -o pipefail
model=${1:-resnet50}
out=gpurun_out/conv1x1_thresholds_$model.txt; : > $out
d=$(mktemp -d /tmp/thr.XXXX)
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$d/t.csv PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
DTF_CONV1X1_MM_FWD_MIN_CIN=32 DTF_CONV1X1_MM_DGRAD_MIN_CIN=32 PYTORCH_TUNABLEOP_TUNING=1 timeout -k 10 600 python -u bench.py --model $model --steps 2 --warmup 2 >/dev/null 2>&1 || exit 1
export PYTORCH_TUNABLEOP_TUNING=0
run() { echo "$* $(env "$@" timeout -k 10 200 python -u bench.py --model $model --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")" | tee -a $out; }
for rep in 1 2; do
run DTF_CONV1X1_MM_FWD_MIN_CIN=1024
run DTF_CONV1X1_MM_FWD_MIN_CIN=64
run DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=256
run DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=128
run DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=512
run DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_DGRAD_MIN_CIN=64
done
cp $d/t0.csv gpurun_out/conv1x1_all_shapes_$model.csv
