#!/bin/bash
# AsyncPS with hook-driven packs / applies: GPU tests + one-worker step time with and without the hooks
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -u -m pytest tests/test_distributed_gpu.py tests/test_multigpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "async" > gpurun_out/r3x_pytest.log 2>&1 || { tail -30 gpurun_out/r3x_pytest.log; exit 1; }
tail -2 gpurun_out/r3x_pytest.log
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 --update-mode async 2>gpurun_out/r3x.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), c.get('update_mode'))")" | tee -a gpurun_out/r3x_matrix.txt; }
: > gpurun_out/r3x_matrix.txt
run async_hooks DTF_FORCE_COMM=1 && run async_nohooks DTF_FORCE_COMM=1 DTF_ASYNC_HOOKS=0 && run async_hooks2 DTF_FORCE_COMM=1 && run async_nohooks2 DTF_FORCE_COMM=1 DTF_ASYNC_HOOKS=0 || { tail gpurun_out/r3x.err; exit 1; }
