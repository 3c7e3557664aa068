#!/bin/bash
# round 2, call E: K4 reorder check, submit->step-1 with the warm pool, config #5 (Inception sync / stale / async)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "embedding or wide_deep" 2>&1 | tail -2 \
 && timeout -k 10 400 python -u scripts/hbm_embedding_bench.py --out gpurun_out/hbm_embedding_cubic.json 2>&1 | grep -E "step_ms|\"ms\"|TBps" \
 && timeout -k 10 300 python -u scripts/submit_latency.py --model resnet50 --steps 40 --batch 256 --warm 2 --out gpurun_out/submit_resnet50_warm.json 2>&1 | grep -E "first step|warm pool" \
 && timeout -k 10 300 python -u scripts/submit_latency.py --model resnet50 --steps 40 --batch 256 --warm 0 --out gpurun_out/submit_resnet50_cold.json 2>&1 | grep -E "first step" \
 && for mode in sync stale async; do DTF_FORCE_COMM=1 timeout -k 10 300 python -u bench.py --model inception_v3 --steps 30 --warmup 8 --update-mode $mode 2>gpurun_out/c5_bench_$mode.err > gpurun_out/c5_bench_$mode.json || exit 1; done \
 && DTF_FORCE_COMM=1 timeout -k 10 300 python -u bench.py --model inception_v3 --steps 30 --warmup 8 --update-mode stale --staleness 2 2>/dev/null > gpurun_out/c5_bench_stale2.json \
 && for mode in sync async; do DTF_FORCE_COMM=1 timeout -k 10 400 python -u scripts/submit_latency.py --model inception_v3 --steps 60 --batch 256 --update-mode $mode --warm 2 --out gpurun_out/c5_rest_$mode.json 2>&1 | grep -E "first step" || exit 1; done \
 && DTF_FORCE_COMM=1 timeout -k 10 400 python -u scripts/submit_latency.py --model inception_v3 --steps 2000 --batch 256 --update-mode async --fault hang:worker=0,step=20 --cancel-after 4 --warm 2 --out gpurun_out/c5_rest_async_hang_cancel.json 2>&1 | grep -E "first step|cancel"
