#!/bin/bash
# round 2, call A: new 1-rank-RCCL tests + graph capture with live collectives
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_distributed_gpu.py tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "rccl or route or sumsq or sharded_ps or wide_deep or bert" 2>&1 | tee gpurun_out/r2a_pytest.log | tail -15 \
 && timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 2>gpurun_out/r2a_eager.err | tee gpurun_out/r2a_eager.json \
 && timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --graph 1 2>gpurun_out/r2a_graph.err | tee gpurun_out/r2a_graph.json \
 && DTF_FORCE_COMM=1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --graph 1 2>gpurun_out/r2a_graph_rccl.err | tee gpurun_out/r2a_graph_rccl.json \
 && DTF_FORCE_COMM=1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --model bert 2>gpurun_out/r2a_bert_rccl.err | tee gpurun_out/r2a_bert_rccl.json \
 && timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --model bert 2>gpurun_out/r2a_bert.err | tee gpurun_out/r2a_bert.json \
 && DTF_FORCE_COMM=1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --model bert --graph 1 2>gpurun_out/r2a_bert_graph_rccl.err | tee gpurun_out/r2a_bert_graph_rccl.json
