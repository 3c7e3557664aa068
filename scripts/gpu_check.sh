#!/bin/bash
# GPU check: kernel numerics, smoke, short bench.  Each step has its own time limit and the
# chain stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tee gpurun_out/pytest_gpu.log | tail -15 \
 && timeout -k 10 300 python -u __graft_entry__.py smoke 2>&1 | tee gpurun_out/smoke.log | tail -2 \
 && timeout -k 10 600 python -u bench.py --steps 20 --warmup 5 2>&1 | tee gpurun_out/bench.log | tail -2
