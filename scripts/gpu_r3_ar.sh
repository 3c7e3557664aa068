#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ar.txt
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 400 python -u bench.py --steps 40 --warmup 8 $EXTRA 2>gpurun_out/r3ar.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['ms_per_step'], 'first', c['first_loss'], 'final_loss', repr(c['final_loss']), c['update_mode'], c['comm_backend'], c['push'])")" | tee -a gpurun_out/r3ar.txt; }
EXTRA=""
run sync_plain DTF_NOOP=1 && run sync_live DTF_FORCE_COMM=1 && run sync_native DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native || { tail gpurun_out/r3ar.err; exit 1; }
EXTRA="--push all_to_all"
run sync_live_a2a DTF_FORCE_COMM=1 || { tail gpurun_out/r3ar.err; exit 1; }
EXTRA="--update-mode async"
run async_live DTF_FORCE_COMM=1 && run async_plain DTF_NOOP=1 || { tail gpurun_out/r3ar.err; exit 1; }
EXTRA="--num-ps 1"
run sync_live_ps1 DTF_FORCE_COMM=1 || { tail gpurun_out/r3ar.err; exit 1; }
