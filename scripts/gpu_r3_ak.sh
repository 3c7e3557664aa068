#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ak.txt
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 400 python -u bench.py --steps 40 --warmup 8 2>gpurun_out/r3ak.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['ms_per_step'], 'final_loss', repr(c['final_loss']))")" | tee -a gpurun_out/r3ak.txt; }
run plain DTF_NOOP=1 && run live DTF_FORCE_COMM=1 && run live_noprefetch DTF_FORCE_COMM=1 DTF_PULL_PREFETCH=0 && \
run live_nowgradstream DTF_FORCE_COMM=1 DTF_WGRAD_STREAM=0 && run plain_nowgradstream DTF_WGRAD_STREAM=0 && \
run live_fp32wire DTF_FORCE_COMM=1 DTF_BENCH_EXTRA=1 || { tail gpurun_out/r3ak.err; exit 1; }
