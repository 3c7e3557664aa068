#!/bin/bash
# Offline GEMM selection for the plain (library) GEMMs: PyTorch TunableOp times every hipBLASLt / rocBLAS solution for
# each GEMM shape the model issues and records the winner in a CSV; later runs only read the CSV.
# usage: gpu_tunableop.sh MODEL   -> gpurun_out/tunableop/<model>.csv + A/B numbers
set -o pipefail
m=${1:-bert}; out=gpurun_out/tunableop; mkdir -p $out
base() { timeout -k 10 300 python -u bench.py --model $m --steps 20 --warmup 6 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; }
echo "$m untuned: $(base)" | tee $out/$m.txt
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=$PWD/$out/$m.csv PYTORCH_TUNABLEOP_VERBOSE=0
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
s=$(date +%s)
PYTORCH_TUNABLEOP_TUNING=1 timeout -k 10 900 python -u bench.py --model $m --steps 2 --warmup 2 >/dev/null 2>$out/$m.tune.err || { tail -5 $out/$m.tune.err; exit 1; }
echo "tuning took $(( $(date +%s) - s )) s; $(ls $out | grep -c csv) csv file(s), $(cat $out/$m*.csv | wc -l) lines" | tee -a $out/$m.txt
export PYTORCH_TUNABLEOP_TUNING=0
echo "$m tuned: $(base)" | tee -a $out/$m.txt
echo "$m tuned: $(base)" | tee -a $out/$m.txt
