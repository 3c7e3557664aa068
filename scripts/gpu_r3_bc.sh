#!/bin/bash
# pipelined execution == serialised execution, bit for bit, with a different batch and learning rate every step?
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3bc.txt
run() { echo "--- $*" | tee -a gpurun_out/r3bc.txt; env "$@" 2>gpurun_out/r3bc.err | tail -2 | cut -c1-260 | tee -a gpurun_out/r3bc.txt; }
for m in "resnet50 64 12" "inception_v3 32 8" "bert 16 8" "wide_deep 65536 10"; do
  run DTF_X=1 timeout -k 10 300 python -u scripts/serial_vs_pipelined.py $m || { tail gpurun_out/r3bc.err; exit 1; }
  run DTF_SERIAL=1 DTF_WGRAD_STREAM=0 DTF_PULL_PREFETCH=0 AMD_SERIALIZE_KERNEL=3 timeout -k 10 400 python -u scripts/serial_vs_pipelined.py $m || { tail gpurun_out/r3bc.err; exit 1; }
done
python3 - <<'PY'
import re
lines = open("gpurun_out/r3bc.txt").read().splitlines()
hs = [l.split(": ")[1] for l in lines if l.startswith("state hash")]
names = [l.split()[0] for l in lines if " batch " in l and "losses" in l]
for i in range(0, len(hs), 2):
    print(names[i], "pipelined == serialised:", hs[i] == hs[i + 1])
PY
