#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
DTF_DETERMINISTIC=1 timeout -k 10 300 python scripts/determinism_check.py 256 2>&1 | grep -E "^iteration|^loss" | cut -c1-330 | tee gpurun_out/r3af.txt
for v in 0 1; do echo "DTF_DETERMINISTIC=$v $(DTF_DETERMINISTIC=$v timeout -k 10 300 python -u bench.py --steps 30 --warmup 8 2>gpurun_out/r3af.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")" | tee -a gpurun_out/r3af.txt; done
