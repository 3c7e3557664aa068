#!/bin/bash
# whole-training reproducibility through the PS pipeline: same final loss run to run?
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ai.txt
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 400 python -u bench.py --steps 40 --warmup 8 $EXTRA 2>gpurun_out/r3ai.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'final_loss', repr(c['final_loss']), c['comm_backend'], c['push'])")" | tee -a gpurun_out/r3ai.txt; }
EXTRA=""
run plain DTF_NOOP=1 && run plain DTF_NOOP=1 && run live DTF_FORCE_COMM=1 && run live DTF_FORCE_COMM=1 && \
run native DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native && run native DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native || { tail gpurun_out/r3ai.err; exit 1; }
EXTRA="--push all_to_all"
run live_a2a DTF_FORCE_COMM=1 && run live_a2a DTF_FORCE_COMM=1 || { tail gpurun_out/r3ai.err; exit 1; }
EXTRA="--model bert"
run bert_live DTF_FORCE_COMM=1 && run bert_live DTF_FORCE_COMM=1 && run bert_plain DTF_NOOP=1 || { tail gpurun_out/r3ai.err; exit 1; }
EXTRA="--update-mode stale"
run stale_live DTF_FORCE_COMM=1 && run stale_live DTF_FORCE_COMM=1 || { tail gpurun_out/r3ai.err; exit 1; }
