#!/bin/bash
# round 3, call F: K19 numerics + bench, then the ResNet-50 bench A/B with and without it
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_stem_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k k19 > gpurun_out/r3f_pytest.log 2>&1 || { tail -30 gpurun_out/r3f_pytest.log; exit 1; }
tail -2 gpurun_out/r3f_pytest.log
timeout -k 10 300 python scripts/wrw1x1_stream_bench.py > gpurun_out/r3f_k19_wrw1x1.txt 2> gpurun_out/r3f_k19.err || { tail -20 gpurun_out/r3f_k19.err; exit 1; }
cat gpurun_out/r3f_k19_wrw1x1.txt
for v in 1 0 1 0; do
  DTF_CONV1X1_WRW_STREAM=$v timeout -k 10 300 python bench.py --steps 30 --warmup 10 --graph 0 2> gpurun_out/r3f_bench_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K19=$v', d['value'], d['ms_per_step'], d['config']['final_loss'])" || { tail gpurun_out/r3f_bench_$v.err; exit 1; }
done | tee gpurun_out/r3f_ab.txt
