#!/bin/bash
# narrow 1x1 weight gradients: MIOpen vs tuned library GEMM
set -o pipefail
model=${1:-resnet50}
out=gpurun_out/wrw_lib_$model.txt; : > $out
d=$(mktemp -d /tmp/thr.XXXX)
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$d/t.csv PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
cp distributed_tensorflow_amd/gemm_db/gfx950_tunableop.csv $d/t0.csv
DTF_CONV1X1_MM_TUNED=0 DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=256 DTF_CONV1X1_MM_DGRAD_MIN_CIN=64 DTF_CONV1X1_WRW_LIB_ALL=1 PYTORCH_TUNABLEOP_TUNING=1 timeout -k 10 600 python -u bench.py --model $model --steps 2 --warmup 2 >/dev/null 2>$out.err || { tail -5 $out.err; exit 1; }
export PYTORCH_TUNABLEOP_TUNING=0
run() { echo "$* $(env "$@" timeout -k 10 200 python -u bench.py --model $model --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])")" | tee -a $out; }
for rep in 1 2; do
run DTF_CONV1X1_MM_TUNED=0 DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=256 DTF_CONV1X1_MM_DGRAD_MIN_CIN=64 DTF_CONV1X1_WRW_LIB_ALL=0
run DTF_CONV1X1_MM_TUNED=0 DTF_CONV1X1_MM_FWD_MIN_CIN=64 DTF_CONV1X1_MM_FWD_SKIP=256 DTF_CONV1X1_MM_DGRAD_MIN_CIN=64 DTF_CONV1X1_WRW_LIB_ALL=1
done
cp $d/t0.csv gpurun_out/wrw_lib_shapes_$model.csv
grep "_NT," gpurun_out/wrw_lib_shapes_$model.csv | cut -d, -f2-4
