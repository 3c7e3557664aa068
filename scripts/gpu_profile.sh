#!/bin/bash
# rocprofv3 kernel-trace of the bench -> gpurun_out/prof ; summary -> gpurun_out/prof/summary.txt
set -o pipefail
mkdir -p gpurun_out/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python3 $R/bench.py --steps 4 --warmup 3 "$@" 2>&1 | grep -v "^[EW]2026" | tail -3
cd $R && python3 scripts/summarize_trace.py gpurun_out/prof/trace_kernel_trace.csv > gpurun_out/prof/summary.txt; sed -n 1,40p gpurun_out/prof/summary.txt
