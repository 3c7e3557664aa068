#!/bin/bash
set -o pipefail
out=gpurun_out/conv1x1_thresholds_more.txt; : > $out
run() { m=$1; shift; echo "$m $* $(env "$@" timeout -k 10 200 python -u bench.py --model $m --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")" | tee -a $out; }
# 1. untuned library: is the low threshold still right?
for rep in 1 2; do
run resnet50 DTF_GEMM_TUNING=0 DTF_CONV1X1_MM_FWD_MIN_CIN=1024
run resnet50 DTF_GEMM_TUNING=0 DTF_CONV1X1_MM_FWD_MIN_CIN=64
run resnet50 DTF_GEMM_TUNING=0 DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=256
done
# 2. Inception-v3 with all shapes tuned
d=$(mktemp -d /tmp/thr.XXXX)
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$d/t.csv PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
DTF_CONV1X1_MM_FWD_MIN_CIN=32 DTF_CONV1X1_MM_DGRAD_MIN_CIN=32 DTF_INCEPTION_HEADS_MM=1 PYTORCH_TUNABLEOP_TUNING=1 timeout -k 10 600 python -u bench.py --model inception_v3 --steps 2 --warmup 2 >/dev/null 2>&1 || exit 1
export PYTORCH_TUNABLEOP_TUNING=0
for rep in 1 2; do
run inception_v3 DTF_CONV1X1_MM_FWD_MIN_CIN=1024
run inception_v3 DTF_CONV1X1_MM_FWD_MIN_CIN=64
run inception_v3 DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_INCEPTION_HEADS_MM=1
run inception_v3 DTF_CONV1X1_MM_FWD_MIN_CIN=1024 DTF_INCEPTION_HEADS_MM=1
run inception_v3 DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_DGRAD_MIN_CIN=32
done
cp $d/t0.csv gpurun_out/conv1x1_all_shapes_inception_v3.csv
