#!/bin/bash
# live-comm path: do the side stream and the compute stream share a hardware queue?  (4 HW queues per process by default)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 2>gpurun_out/r3p.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), c.get('comm_backend'))")" | tee -a gpurun_out/r3p_matrix.txt; }
: > gpurun_out/r3p_matrix.txt
run live_q4 DTF_FORCE_COMM=1 && run live_q8 DTF_FORCE_COMM=1 GPU_MAX_HW_QUEUES=8 && run plain_q8 GPU_MAX_HW_QUEUES=8 && \
run live_q6 DTF_FORCE_COMM=1 GPU_MAX_HW_QUEUES=6 && run live_native_q8 DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native GPU_MAX_HW_QUEUES=8 && \
run live_q8_b DTF_FORCE_COMM=1 GPU_MAX_HW_QUEUES=8 && run plain_q4 DTF_NOOP=1 || { tail gpurun_out/r3p.err; exit 1; }
