#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 1 0 1 0; do
  DTF_COMPUTE_PRIORITY=$v timeout -k 10 300 python bench.py --steps 30 --warmup 10 2> gpurun_out/r3l_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('PRIORITY=$v', d['value'], d['ms_per_step'], 'host', c['host_enqueue_ms_per_step'], 'graph', c['hipgraph'])" || { tail gpurun_out/r3l_bench.err; exit 1; }
done
scripts/gpu_profile.sh > gpurun_out/r3l_prof.txt 2>&1; sed -n 1,45p gpurun_out/prof/summary.txt
