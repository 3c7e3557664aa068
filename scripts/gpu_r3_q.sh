#!/bin/bash
# the package now asks for 8 hardware queues itself: live 1-rank RCCL vs no communicator, nothing set on the command line
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 2>gpurun_out/r3q.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), c.get('comm_backend'), 'queues', c.get('gpu_max_hw_queues'))")" | tee -a gpurun_out/r3q_matrix.txt; }
: > gpurun_out/r3q_matrix.txt
echo "box GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-unset}" | tee -a gpurun_out/r3q_matrix.txt
run live DTF_FORCE_COMM=1 && run plain DTF_NOOP=1 && run live_keep4 DTF_FORCE_COMM=1 DTF_KEEP_HW_QUEUES=1 && run live_native DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native || { tail gpurun_out/r3q.err; exit 1; }
