#!/bin/bash
# hipGraph replay with 8 hardware queues: does the captured side-stream branch overlap now?
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; g=$2; shift 2; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --graph $g 2>gpurun_out/r3s.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), c.get('comm_backend'), 'graph', c.get('hipgraph'), 'queues', c.get('gpu_max_hw_queues'), c.get('hipgraph_trial'))")" | tee -a gpurun_out/r3s_matrix.txt; }
: > gpurun_out/r3s_matrix.txt
run eager 0 DTF_NOOP=1 && run graph 1 DTF_NOOP=1 && run graph_q16 1 GPU_MAX_HW_QUEUES=16 && run graph_q4 1 DTF_KEEP_HW_QUEUES=1 && \
run live_eager 0 DTF_FORCE_COMM=1 && run live_graph 1 DTF_FORCE_COMM=1 && run graph_nowgrad 1 DTF_WGRAD_STREAM=0 || { tail gpurun_out/r3s.err; exit 1; }
