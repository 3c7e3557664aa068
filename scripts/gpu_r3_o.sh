#!/bin/bash
# why is the live 1-rank RCCL path slower than the group-less path since the side stream?  A/B matrix + trace
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 2>gpurun_out/r3o.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), c.get('comm_backend'))")" | tee -a gpurun_out/r3o_matrix.txt; }
: > gpurun_out/r3o_matrix.txt
run plain DTF_NOOP=1 && run live DTF_FORCE_COMM=1 && run live_noprefetch DTF_FORCE_COMM=1 DTF_PULL_PREFETCH=0 && \
run live_nowgradstream DTF_FORCE_COMM=1 DTF_WGRAD_STREAM=0 && run plain_nowgradstream DTF_WGRAD_STREAM=0 && \
run live_native DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native && run live2 DTF_FORCE_COMM=1 || { tail gpurun_out/r3o.err; exit 1; }
export DTF_FORCE_COMM=1
scripts/gpu_profile.sh > gpurun_out/r3o_prof.txt 2>&1; cp gpurun_out/prof/summary.txt gpurun_out/r3o_live_summary.txt; sed -n 1,45p gpurun_out/prof/summary.txt
