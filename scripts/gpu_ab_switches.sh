#!/bin/bash
# A/B of routing / tuning switches on ONE box: baseline first, last and between the groups
run() { echo "$1 $(env $1 timeout -k 10 200 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")"; }
run DTF_NOOP=1
for e in "$@"; do run $e || exit 1; done
run DTF_NOOP=1
