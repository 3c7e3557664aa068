#!/bin/bash
# after restore_hw_queues(): graph runs get HIP's default queue count back, eager runs keep 8
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; g=$2; shift 2; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --graph $g 2>gpurun_out/r3u.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), c.get('comm_backend'), 'graph', c.get('hipgraph'), 'queues', c.get('gpu_max_hw_queues'), c.get('hipgraph_trial'))")" | tee -a gpurun_out/r3u_matrix.txt; }
: > gpurun_out/r3u_matrix.txt
run graph 1 DTF_NOOP=1 && run live_graph 1 DTF_FORCE_COMM=1 && run live_eager 0 DTF_FORCE_COMM=1 && run trial -1 DTF_NOOP=1 || { tail gpurun_out/r3u.err; exit 1; }
timeout -k 10 600 python -u -m pytest tests/test_multigpu.py tests/test_distributed_gpu.py tests/test_rccl_native_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3u_pytest.log 2>&1 || { tail -30 gpurun_out/r3u_pytest.log; exit 1; }
tail -3 gpurun_out/r3u_pytest.log
