#!/bin/bash
# kernel-trace summaries of the model families -> gpurun_out/prof_<model>.txt
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
for m in "$@"; do
  rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o trace -- python3 $R/bench.py --model $m --steps 4 --warmup 3 2>&1 | grep -v "^[EW]2026" | tail -1 | cut -c1-200 || exit 1
  cd $R && python3 scripts/summarize_trace.py gpurun_out/prof/trace_kernel_trace.csv > gpurun_out/prof_$m.txt || exit 1
  echo "== $m"; sed -n 1,3p gpurun_out/prof_$m.txt
done
rm -rf $R/gpurun_out/prof
