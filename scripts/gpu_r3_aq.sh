#!/bin/bash
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 1 0 1 0; do echo "inception det=$v $(DTF_DETERMINISTIC=$v timeout -k 10 400 python -u bench.py --model inception_v3 --steps 30 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")"; done
