#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 32 48 96 128; do echo "== fwd batch $b"; timeout -k 10 200 python scripts/fwd_repro_check.py $b 2>&1 | grep -E "^losses|^modules" | cut -c1-400; done
for b in 32 128; do echo "== grads batch $b"; timeout -k 10 200 python scripts/determinism_check.py $b 2>&1 | grep -E "^iteration|^loss" | cut -c1-300; done
for i in 1 2; do echo "bench $(timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")"; done
