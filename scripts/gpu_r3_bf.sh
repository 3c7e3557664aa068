#!/bin/bash
# emb_chunk_partial with four gathers in flight: tests, Wide&Deep A/B is not possible in one build, so before/after by kernel time
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q --timeout 120 --timeout-method thread -k "embedding or wide_deep or segment" 2>&1 | tail -3 || exit 1
for i in 1 2 3; do timeout -k 10 300 python -u bench.py --model wide_deep --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wide_deep', j['value'], j['ms_per_step'], j['config']['final_loss'])"; done | tee gpurun_out/r3bf.txt
bash scripts/gpu_profile_all.sh wide_deep > /dev/null 2>&1; grep -E "emb_chunk_partial|emb_segment_apply|one training step" gpurun_out/prof_wide_deep.txt | tee -a gpurun_out/r3bf.txt
