#!/bin/bash
# rocprofv3 kernel-trace of the bench
set -o pipefail
mkdir -p gpurun_out/prof
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o rn50 -- python3 $R/bench.py --steps 5 --warmup 3 --miopen-find 0 2>&1 | tail -5
ls -R $R/gpurun_out/prof | head -20
