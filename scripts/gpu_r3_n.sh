#!/bin/bash
# round-3 refresh: all-model table, per-model step anatomies, HBM-scale embedding step restricted to the step window
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
scripts/gpu_all_models.sh || exit 1
scripts/gpu_profile_all.sh inception_v3 bert wide_deep || exit 1
mkdir -p $R/gpurun_out/prof_hbm
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_hbm -o trace -- python3 $R/scripts/hbm_embedding_bench.py --steps 6 --out $R/gpurun_out/r3_hbm_embedding.json 2>&1 | grep -v "^[EW]2026" | tail -5 || exit 1
cd $R && python3 scripts/summarize_trace.py gpurun_out/prof_hbm/trace_kernel_trace.csv > gpurun_out/prof_hbm_embedding.txt && sed -n 1,25p gpurun_out/prof_hbm_embedding.txt
rm -rf $R/gpurun_out/prof_hbm
