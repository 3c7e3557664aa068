#!/bin/bash
# consistency matrix: the same training through different data-plane paths must end at the same loss (one rank)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3an.txt
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 400 python -u bench.py --steps 40 --warmup 8 $EXTRA 2>gpurun_out/r3an.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['ms_per_step'], 'first', c['first_loss'], 'final_loss', repr(c['final_loss']), c['update_mode'], c['comm_backend'], 'graph', c['hipgraph'])")" | tee -a gpurun_out/r3an.txt; }
EXTRA="--model bert"
run bert_plain DTF_NOOP=1 && run bert_live DTF_FORCE_COMM=1 && run bert_live_noprefetch DTF_FORCE_COMM=1 DTF_PULL_PREFETCH=0 || { tail gpurun_out/r3an.err; exit 1; }
EXTRA="--model inception_v3"
run inc_plain DTF_NOOP=1 && run inc_live DTF_FORCE_COMM=1 || { tail gpurun_out/r3an.err; exit 1; }
EXTRA="--model wide_deep"
run wd_plain DTF_NOOP=1 && run wd_plain2 DTF_NOOP=1 && run wd_live DTF_FORCE_COMM=1 || { tail gpurun_out/r3an.err; exit 1; }
EXTRA="--update-mode stale"
run stale_plain DTF_NOOP=1 && run stale_live DTF_FORCE_COMM=1 && run stale_live2 DTF_FORCE_COMM=1 || { tail gpurun_out/r3an.err; exit 1; }
EXTRA="--update-mode async"
run async_live DTF_FORCE_COMM=1 && run async_live_nohooks DTF_FORCE_COMM=1 DTF_ASYNC_HOOKS=0 || { tail gpurun_out/r3an.err; exit 1; }
EXTRA="--graph 1"
run graph_plain DTF_NOOP=1 && run graph_live DTF_FORCE_COMM=1 || { tail gpurun_out/r3an.err; exit 1; }
