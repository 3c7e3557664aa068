#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 8 2>gpurun_out/r3k_bench_default.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('default', d['value'], d['ms_per_step'], 'graph', c['hipgraph'], c['hipgraph_trial'], 'host', c['host_enqueue_ms_per_step'])" || { tail gpurun_out/r3k_bench_default.err; exit 1; }
done
for m in resnet50 resnet152; do
  timeout -k 10 300 python bench.py --model $m --steps 30 --warmup 10 --graph 0 2> gpurun_out/r3k_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$m eager', d['value'], d['ms_per_step'], 'host', c['host_enqueue_ms_per_step'], c['host_loop_ms_per_step'])" || { tail gpurun_out/r3k_bench.err; exit 1; }
done
