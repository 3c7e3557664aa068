#!/bin/bash
# round 2, call B: graph-vs-eager bitwise with live RCCL, one-sided async PS over HIP IPC, BERT graph bench
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_distributed_gpu.py -m gpu -x -q --timeout 240 --timeout-method thread 2>&1 | tee gpurun_out/r2b_pytest.log | tail -15 \
 && DTF_FORCE_COMM=1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --model bert --graph 1 2>gpurun_out/r2b_bert_graph_rccl.err | tee gpurun_out/r2b_bert_graph_rccl.json \
 && DTF_FORCE_COMM=1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --graph 1 2>gpurun_out/r2b_rn_graph_rccl.err | tee gpurun_out/r2b_rn_graph_rccl.json \
 && timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 --update-mode async 2>gpurun_out/r2b_rn_async.err | tee gpurun_out/r2b_rn_async.json
