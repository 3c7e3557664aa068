#!/bin/bash
# fused BN backward finalize: numerics + reproducibility tests, then the whole-step A/B
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -u -m pytest tests/test_kernels_gpu.py tests/test_stem_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread \
  -k "bn or stem or pool or resnet" > gpurun_out/r3m_pytest.log 2>&1 || { tail -30 gpurun_out/r3m_pytest.log; exit 1; }
tail -3 gpurun_out/r3m_pytest.log
for v in 1 0 1 0; do
  DTF_BN_FUSED_FINALIZE=$v timeout -k 10 300 python bench.py --steps 30 --warmup 10 2> gpurun_out/r3m_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('FUSED_FINALIZE=$v', d['value'], d['ms_per_step'], 'host', c['host_enqueue_ms_per_step'], 'loss', c.get('first_loss'))" || { tail gpurun_out/r3m_bench.err; exit 1; }
done
