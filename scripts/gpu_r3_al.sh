#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -u -m pytest tests/test_distributed_gpu.py tests/test_multigpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3al_pytest.log 2>&1 || { tail -30 gpurun_out/r3al_pytest.log; exit 1; }
tail -2 gpurun_out/r3al_pytest.log
: > gpurun_out/r3al.txt
run() { name=$1; shift; echo "$name $(env "$@" timeout -k 10 400 python -u bench.py --steps 40 --warmup 8 $EXTRA 2>gpurun_out/r3al.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print(j['ms_per_step'], 'first', c['first_loss'], 'final_loss', repr(c['final_loss']))")" | tee -a gpurun_out/r3al.txt; }
EXTRA="--update-mode stale"
run stale_live DTF_FORCE_COMM=1 && run stale_live DTF_FORCE_COMM=1 && run stale_plain DTF_NOOP=1 || { tail gpurun_out/r3al.err; exit 1; }
EXTRA="--update-mode stale --lr-warmup-steps 0"
run stale_live_constlr DTF_FORCE_COMM=1 || { tail gpurun_out/r3al.err; exit 1; }
EXTRA="--model bert"
run bert_live DTF_FORCE_COMM=1 && run bert_plain DTF_NOOP=1 || { tail gpurun_out/r3al.err; exit 1; }
