#!/bin/bash
set -o pipefail
b() { env "$1" timeout -k 10 200 python bench.py --steps 40 --warmup 15 "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$*', r['value'], r['ms_per_step'])"; }
b DTF_CONV_WRW_DIRECT=0 && b DTF_CONV_WRW_DIRECT=1 && b DTF_CONV_WRW_DIRECT=0 && b DTF_CONV_WRW_DIRECT=1 && b DTF_CONV_WRW_DIRECT=0 --model inception_v3 && b DTF_CONV_WRW_DIRECT=1 --model inception_v3
