#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/prof_hbm
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py tests/test_distributed_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "embedding or wide_deep or segment or route" 2>&1 | tail -4 \
 && timeout -k 10 500 python -u scripts/hbm_embedding_bench.py --out gpurun_out/hbm_embedding_cubic.json 2>&1 | tee gpurun_out/hbm_embedding_cubic.log | grep -E "hbm\]|step_ms|examples|\"ms\"|TBps" \
 && DTF_FORCE_COMM=1 timeout -k 10 500 python -u scripts/hbm_embedding_bench.py --skew zipf --out gpurun_out/hbm_embedding_zipf_rccl1.json 2>&1 | tee gpurun_out/hbm_embedding_zipf.log | grep -E "hbm\]|step_ms|examples|\"ms\"|TBps" \
 && timeout -k 10 300 python -u bench.py --model wide_deep --steps 30 --warmup 5 2>/dev/null | tee gpurun_out/r2d_wide_deep.json | cut -c1-200 \
 && cd /tmp && export TMPDIR=/tmp \
 && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hbm -o trace -- python3 $R/scripts/hbm_embedding_bench.py --steps 4 --out $R/gpurun_out/hbm_embedding_prof.json 2>&1 | grep -v "^[EW]2026" | tail -1
