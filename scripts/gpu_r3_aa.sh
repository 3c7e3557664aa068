#!/bin/bash
# restricted-bit id sort for K4: numerics, embedding tests, Wide&Deep A/B (small tables + HBM scale)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -u -m pytest tests/test_kernels_gpu.py tests/test_distributed_gpu.py tests/test_multigpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "sort or wide or embedding or routed or sparse" > gpurun_out/r3aa_pytest.log 2>&1 || { tail -30 gpurun_out/r3aa_pytest.log; exit 1; }
tail -2 gpurun_out/r3aa_pytest.log
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 300 python -u bench.py --model wide_deep --steps 50 --warmup 10 2>gpurun_out/r3aa.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'), 'loss', c.get('first_loss'), c.get('final_loss'))")" | tee -a gpurun_out/r3aa_matrix.txt; }
: > gpurun_out/r3aa_matrix.txt
run sort_kernel DTF_NOOP=1 && run torch_sort DTF_EMB_SORT_KERNEL=0 && run sort_kernel DTF_NOOP=1 && run torch_sort DTF_EMB_SORT_KERNEL=0 || { tail gpurun_out/r3aa.err; exit 1; }
timeout -k 10 500 python -u scripts/hbm_embedding_bench.py --steps 20 --out gpurun_out/r3aa_hbm.json > gpurun_out/r3aa_hbm.log 2>&1 || { tail gpurun_out/r3aa_hbm.log; exit 1; }
grep -E "step_ms|examples_per_sec|\"ms\"|effective" gpurun_out/r3aa_hbm.json
