#!/bin/bash
# round 3, call B: bench with the new defaults (graph auto, lr warm-up), eager A/B, W&D rccl test, step profile
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_distributed_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k wide_deep > gpurun_out/r3b_pytest.log 2>&1 || { tail -20 gpurun_out/r3b_pytest.log; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 8 > gpurun_out/r3b_bench_graph_auto.json 2> gpurun_out/r3b_bench.err && cat gpurun_out/r3b_bench_graph_auto.json || { tail -20 gpurun_out/r3b_bench.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 8 --graph 0 > gpurun_out/r3b_bench_eager.json 2> gpurun_out/r3b_bench_eager.err && cat gpurun_out/r3b_bench_eager.json || exit 1
timeout -k 10 300 python bench.py --steps 40 --warmup 60 --graph 0 > gpurun_out/r3b_bench_eager_100.json 2> gpurun_out/r3b_bench_eager100.err && cat gpurun_out/r3b_bench_eager_100.json || exit 1
scripts/gpu_profile.sh --graph 0
