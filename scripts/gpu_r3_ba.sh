#!/bin/bash
# Wide&Deep through a live 1-rank group is slower than without one (3.35 vs 2.60 ms): which part?  nccl vs native communicator, route forms
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ba.txt
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 400 python -u bench.py --model wide_deep --steps 60 --warmup 10 $EXTRA 2>gpurun_out/r3ba.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), c.get('host_loop_ms_per_step'), 'final_loss', repr(c['final_loss']), c['comm_backend'])")" | tee -a gpurun_out/r3ba.txt; }
EXTRA=""
run wd_plain DTF_NOOP=1 && run wd_live_nccl DTF_FORCE_COMM=1 && run wd_live_native DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native && run wd_plain DTF_NOOP=1 && run wd_live_nccl_graph DTF_FORCE_COMM=1 || { tail gpurun_out/r3ba.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
DTF_FORCE_COMM=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wd_live -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --model wide_deep --steps 4 --warmup 3 2>&1 | grep -v "^[EW]2026" | tail -1 | cut -c1-150
cd $GRAFT_REPO_ROOT && python3 scripts/summarize_trace.py gpurun_out/prof_wd_live/trace_kernel_trace.csv > gpurun_out/prof_wd_live.txt && sed -n 1,40p gpurun_out/prof_wd_live.txt && rm -rf gpurun_out/prof_wd_live
