#!/bin/bash
# Scaling sweep on a multi-GPU node: bench.py under torchrun at 1/2/4/8 GPUs x {reduce_scatter, all_to_all} x
# bucket {8,16,32,64} MB x graph {0,1}, NCCL_DEBUG=INFO rank-count lines captured -> gpurun_out/scale_sweep.json.
# On a one-GPU box only the N=1 column runs (through a live 1-rank RCCL group).  Arguments go to scale_sweep.py.
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
exec python scripts/scale_sweep.py "$@"
