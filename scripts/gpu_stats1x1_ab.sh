#!/bin/bash
set -o pipefail
b() { DTF_CONV1X1_STATS=$1 DTF_CONV1X1_STATS_MIN_COUT=$2 timeout -k 10 200 python bench.py --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('resnet50 conv1x1_stats=$1 min_cout=$2', r['value'], r['ms_per_step'])"; }
b 0 256 && b 1 256 && b 1 1024 && b 1 512 && b 0 256 && b 1 256
