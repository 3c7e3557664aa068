#!/bin/bash
# round 3, call D: native RCCL communicator tests + host cost + bench on it; adaptive warm hint
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_rccl_native_gpu.py -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/r3d_pytest_native.log 2>&1 || { tail -40 gpurun_out/r3d_pytest_native.log; exit 1; }
tail -3 gpurun_out/r3d_pytest_native.log
timeout -k 10 600 python -u -m pytest tests/test_multigpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k native > gpurun_out/r3d_pytest_native_mg.log 2>&1 || { tail -40 gpurun_out/r3d_pytest_native_mg.log; exit 1; }
tail -3 gpurun_out/r3d_pytest_native_mg.log
timeout -k 10 200 python scripts/comm_host_cost.py > gpurun_out/r3d_comm_host_cost.txt 2> gpurun_out/r3d_comm_host_cost.err || { tail -20 gpurun_out/r3d_comm_host_cost.err; exit 1; }
cat gpurun_out/r3d_comm_host_cost.txt
DTF_FORCE_COMM=1 timeout -k 10 300 python bench.py --steps 20 --warmup 8 --graph 0 > gpurun_out/r3d_bench_c10d_eager.json 2> gpurun_out/r3d_b1.err || { tail gpurun_out/r3d_b1.err; exit 1; }
DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native timeout -k 10 300 python bench.py --steps 20 --warmup 8 --graph 0 > gpurun_out/r3d_bench_native_eager.json 2> gpurun_out/r3d_b2.err || { tail gpurun_out/r3d_b2.err; exit 1; }
DTF_FORCE_COMM=1 DTF_COMM_BACKEND=rccl-native timeout -k 10 300 python bench.py --steps 20 --warmup 8 --graph 1 > gpurun_out/r3d_bench_native_graph.json 2> gpurun_out/r3d_b3.err || { tail gpurun_out/r3d_b3.err; exit 1; }
cat gpurun_out/r3d_bench_c10d_eager.json gpurun_out/r3d_bench_native_eager.json gpurun_out/r3d_bench_native_graph.json | cut -c1-200
