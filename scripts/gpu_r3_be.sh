#!/bin/bash
# pipelined == serialised at the BENCH shapes (tuned library choices, largest tensors): ResNet-50 b256, Inception-v3 b256, BERT b64
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3be.txt
S="DTF_SERIAL=1 DTF_WGRAD_STREAM=0 DTF_PULL_PREFETCH=0 AMD_SERIALIZE_KERNEL=3"
run() { echo "--- $*" | tee -a gpurun_out/r3be.txt; env "$@" 2>gpurun_out/r3be.err | tail -2 | cut -c1-300 | tee -a gpurun_out/r3be.txt; }
for m in "resnet50 256 8" "inception_v3 256 6" "bert 64 6"; do
  run DTF_X=1 timeout -k 10 400 python -u scripts/serial_vs_pipelined.py $m && run $S timeout -k 10 500 python -u scripts/serial_vs_pipelined.py $m || { tail gpurun_out/r3be.err; exit 1; }
done
python3 - <<'PY'
lines = open("gpurun_out/r3be.txt").read().splitlines()
hs = [l.split(": ")[1] for l in lines if l.startswith("state hash")]
names = [l.split()[0] + " b" + l.split()[2] for l in lines if " batch " in l and "losses" in l]
for i in range(0, len(hs), 2):
    print(names[i], "pipelined == serialised:", hs[i] == hs[i + 1])
PY
