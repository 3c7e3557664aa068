#!/bin/bash
# host-side profile (cProfile) of the live-communicator step: where do the 17-19 ms of enqueue time per step go?
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
DTF_FORCE_COMM=1 timeout -k 10 400 python -u -m cProfile -o gpurun_out/r3r_live.pstats bench.py --steps 40 --warmup 8 > gpurun_out/r3r_live.json 2> gpurun_out/r3r.err || { tail gpurun_out/r3r.err; exit 1; }
tail -c 600 gpurun_out/r3r_live.json
python - <<'PY'
import pstats
p = pstats.Stats('gpurun_out/r3r_live.pstats')
p.sort_stats('tottime').print_stats(45)
PY
