#!/bin/bash
# re-time the shipped GEMM choices on aligned operands, then A/B shipped vs re-tuned file through bench.py
set -o pipefail
mkdir -p gpurun_out/tunableop
timeout -k 10 600 python -u scripts/retune_gemms.py > gpurun_out/tunableop/retune.log 2>&1 || { tail -20 gpurun_out/tunableop/retune.log; exit 1; }
tail -3 gpurun_out/tunableop/retune.log
b() { f=$1; shift; DTF_GEMM_DB=$PWD/$f timeout -k 10 200 python bench.py --steps 40 --warmup 15 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$f $*', r['value'], r['ms_per_step'])"; }
S=distributed_tensorflow_amd/gemm_db/gfx950_tunableop.csv; N=gpurun_out/tunableop/merged.csv
b $S --model resnet50 && b $N --model resnet50 && b $S --model resnet50 && b $N --model resnet50 && b $S --model bert && b $N --model bert && b $S --model bert && b $N --model bert
