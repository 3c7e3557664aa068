#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ao.txt
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 400 python -u bench.py --steps 40 --warmup 8 $EXTRA 2>gpurun_out/r3ao.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['ms_per_step'], 'first', c['first_loss'], 'final_loss', repr(c['final_loss']))")" | tee -a gpurun_out/r3ao.txt; }
EXTRA="--model inception_v3"
run inc_plain DTF_NOOP=1 && run inc_plain DTF_NOOP=1 && run inc_live DTF_FORCE_COMM=1 && run inc_live DTF_FORCE_COMM=1 && run inc_live_noprefetch DTF_FORCE_COMM=1 DTF_PULL_PREFETCH=0 || { tail gpurun_out/r3ao.err; exit 1; }
