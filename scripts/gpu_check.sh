#!/bin/bash
# First-contact GPU run: kernel numerics, smoke, short bench.  Each step has its own
# time limit and the chain stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showmeminfo vram 2>/dev/null | head -8 > gpurun_out/rocm_smi.txt
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tee gpurun_out/pytest_gpu.log | tail -40 \
 && timeout -k 10 300 python -u __graft_entry__.py smoke 2>&1 | tee gpurun_out/smoke.log | tail -5 \
 && timeout -k 10 600 python -u bench.py --steps 10 --warmup 5 --miopen-find 0 2>&1 | tee gpurun_out/bench_nofind.log | tail -3
