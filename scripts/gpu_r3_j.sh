#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3j_pytest_gpu.log 2>&1
rc=$?; tail -3 gpurun_out/r3j_pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
for m in resnet50 wide_deep bert resnet152 inception_v3; do
for v in 1 0; do
  DTF_WGRAD_STREAM=$v timeout -k 10 300 python bench.py --model $m --steps 30 --warmup 10 --graph 0 2> gpurun_out/r3j_bench_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$m WGRAD_STREAM=$v', d['value'], d['ms_per_step'], 'host', c['host_enqueue_ms_per_step'], 'loss', c['first_loss'], c['final_loss'])" || { tail gpurun_out/r3j_bench_$v.err; exit 1; }
done; done | tee gpurun_out/r3j_ab.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 8 > gpurun_out/r3j_bench_default.json 2>gpurun_out/r3j_bench_default.err && cat gpurun_out/r3j_bench_default.json
