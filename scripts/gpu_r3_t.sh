#!/bin/bash
# graph replay: 29 ms without the side stream / with a live group -- which factor?
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; g=$2; shift 2; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --graph $g 2>gpurun_out/r3t.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), c.get('comm_backend'), 'graph', c.get('hipgraph'), 'queues', c.get('gpu_max_hw_queues'))")" | tee -a gpurun_out/r3t_matrix.txt; }
: > gpurun_out/r3t_matrix.txt
run graph_nowgrad_q4 1 DTF_WGRAD_STREAM=0 DTF_KEEP_HW_QUEUES=1 && run graph_nowgrad_q8 1 DTF_WGRAD_STREAM=0 && \
run live_graph_q4 1 DTF_FORCE_COMM=1 DTF_KEEP_HW_QUEUES=1 && run graph_nowgrad_nolrwarm 1 DTF_WGRAD_STREAM=0 DTF_BENCH_NO_LR_WARMUP=1 && \
run eager_nowgrad 0 DTF_WGRAD_STREAM=0 || { tail gpurun_out/r3t.err; exit 1; }
