#!/bin/bash
# full GPU suite + refreshed GEMM table + all models
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2g_pytest.log 2>&1 || { tail -40 gpurun_out/r2g_pytest.log; exit 1; }
tail -3 gpurun_out/r2g_pytest.log
timeout -k 10 300 python -u scripts/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1 || { tail gpurun_out/gemm_bench.log; exit 1; }
bash scripts/gpu_all_models.sh
