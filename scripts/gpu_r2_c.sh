#!/bin/bash
# round 2, call C: Wide&Deep at HBM scale (180 GB of table + Adagrad state on one MI355X)
set -o pipefail
mkdir -p gpurun_out/prof_hbm
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
timeout -k 10 500 python -u scripts/hbm_embedding_bench.py --out gpurun_out/hbm_embedding_cubic.json 2>&1 | tee gpurun_out/hbm_embedding_cubic.log | tail -45 \
 && DTF_FORCE_COMM=1 timeout -k 10 500 python -u scripts/hbm_embedding_bench.py --skew zipf --out gpurun_out/hbm_embedding_zipf_rccl1.json 2>&1 | tee gpurun_out/hbm_embedding_zipf.log | tail -45 \
 && cd /tmp && export TMPDIR=/tmp \
 && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hbm -o trace -- python3 $R/scripts/hbm_embedding_bench.py --steps 4 --out $R/gpurun_out/hbm_embedding_prof.json 2>&1 | grep -v "^[EW]2026" | tail -3
