#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3aj.txt
for w in 1 2 3 6; do
for cfg in "DTF_NOOP=1" "DTF_FORCE_COMM=1"; do
echo "warmup=$w $cfg $(env $cfg timeout -k 10 300 python -u bench.py --steps 1 --warmup $w 2>gpurun_out/r3aj.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j['config']; print('first', repr(c['first_loss']), 'final', repr(c['final_loss']))")" | tee -a gpurun_out/r3aj.txt
done; done
