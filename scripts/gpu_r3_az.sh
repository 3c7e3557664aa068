#!/bin/bash
# Inception-v3 in the DEFAULT deterministic mode (K21 for the stem layer): same loss through every data-plane path?
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3az.txt
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 400 python -u bench.py --model inception_v3 --steps 40 --warmup 8 $EXTRA 2>gpurun_out/r3az.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['ms_per_step'], 'first', c['first_loss'], 'final_loss', repr(c['final_loss']), c['update_mode'], c['comm_backend'])")" | tee -a gpurun_out/r3az.txt; }
EXTRA=""
run inc_plain DTF_NOOP=1 && run inc_plain DTF_NOOP=1 && run inc_live DTF_FORCE_COMM=1 && run inc_live_noprefetch DTF_FORCE_COMM=1 DTF_PULL_PREFETCH=0 && run inc_native DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native || { tail gpurun_out/r3az.err; exit 1; }
EXTRA="--update-mode async"
run inc_async DTF_FORCE_COMM=1 || { tail gpurun_out/r3az.err; exit 1; }
