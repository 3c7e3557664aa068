#!/bin/bash
# pipelined == serialised for the other data-plane forms: all-to-all push, native communicator, stale-2, async (ResNet-50, varying batches)
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3bd.txt
S="DTF_SERIAL=1 DTF_WGRAD_STREAM=0 DTF_PULL_PREFETCH=0 AMD_SERIALIZE_KERNEL=3"
run() { echo "--- $*" | tee -a gpurun_out/r3bd.txt; env "$@" timeout -k 10 400 python -u scripts/serial_vs_pipelined.py resnet50 64 10 2>gpurun_out/r3bd.err | tail -2 | cut -c1-300 | tee -a gpurun_out/r3bd.txt; }
run DTF_PUSH=all_to_all && run DTF_PUSH=all_to_all $S && \
run DTF_COMM_BACKEND=rccl-native && run DTF_COMM_BACKEND=rccl-native $S && \
run DTF_UPDATE_MODE=stale DTF_STALENESS=2 && run DTF_UPDATE_MODE=stale DTF_STALENESS=2 $S && \
run DTF_UPDATE_MODE=async && run DTF_UPDATE_MODE=async $S || { tail gpurun_out/r3bd.err; exit 1; }
python3 - <<'PY'
lines = open("gpurun_out/r3bd.txt").read().splitlines()
hs = [l.split(": ")[1] for l in lines if l.startswith("state hash")]
cmds = [l for l in lines if l.startswith("--- ")]
for i in range(0, len(hs), 2):
    print(cmds[i][4:60], "| pipelined == serialised:", hs[i] == hs[i + 1])
print("sync forms all equal:", len({hs[0], hs[2], hs[6]}) == 1, "(all-to-all push, native communicator, async at one rank)")
PY
