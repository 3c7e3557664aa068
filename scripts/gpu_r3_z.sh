#!/bin/bash
# K20 row split / merge: numerics, then Wide&Deep with and without it (small tables and HBM scale)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -u -m pytest tests/test_kernels_gpu.py tests/test_distributed_gpu.py tests/test_multigpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "split or wide or embedding or routed" > gpurun_out/r3z_pytest.log 2>&1 || { tail -30 gpurun_out/r3z_pytest.log; exit 1; }
tail -2 gpurun_out/r3z_pytest.log
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --model wide_deep --steps 50 --warmup 10 2>gpurun_out/r3z.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), 'loss', c.get('first_loss'), c.get('final_loss'))")" | tee -a gpurun_out/r3z_matrix.txt; }
: > gpurun_out/r3z_matrix.txt
run k20 DTF_NOOP=1 && run tensor_ops DTF_EMB_SPLIT_KERNEL=0 && run k20 DTF_NOOP=1 && run tensor_ops DTF_EMB_SPLIT_KERNEL=0 || { tail gpurun_out/r3z.err; exit 1; }
