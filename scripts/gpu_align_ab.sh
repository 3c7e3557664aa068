#!/bin/bash
set -o pipefail
b() { timeout -k 10 200 python bench.py --steps 40 --warmup 15 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$*', r['value'], r['ms_per_step'])"; }
b --model resnet50 && b --model resnet50 && b --model inception_v3 && b --model bert
