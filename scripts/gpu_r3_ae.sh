#!/bin/bash
# after the gemm_tn fix (LDS reads retired before the barrier): numerics + regression test, kernel times, step times
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread -k "gemm_tn or conv_wrw or linear or wrw" > gpurun_out/r3ae_pytest.log 2>&1 || { tail -30 gpurun_out/r3ae_pytest.log; exit 1; }
tail -2 gpurun_out/r3ae_pytest.log
timeout -k 10 300 python scripts/gemm_tn_bench.py 2>/dev/null | tee gpurun_out/r3ae_tn_bench.txt || exit 1
run() { name=$1; shift; echo "$name $(timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 "$@" 2>gpurun_out/r3ae.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['value'], j['ms_per_step'], 'host', c.get('host_enqueue_ms_per_step'))")" | tee -a gpurun_out/r3ae_matrix.txt; }
: > gpurun_out/r3ae_matrix.txt
run resnet50 && run resnet50 && run bert --model bert && run inception_v3 --model inception_v3 && run wide_deep --model wide_deep || { tail gpurun_out/r3ae.err; exit 1; }
