#!/bin/bash
# strict deterministic mode on Inception-v3 after it took over the untuned / strided products: reproducibility, routes, cost
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3ax.txt
for b in 256 48; do
  echo "== inception_v3 batch $b DTF_DETERMINISTIC=2" | tee -a gpurun_out/r3ax.txt
  DTF_DETERMINISTIC=2 timeout -k 10 400 python -u scripts/determinism_check.py $b inception_v3 2>gpurun_out/r3ax.err | tail -2 | cut -c1-300 | tee -a gpurun_out/r3ax.txt || { tail gpurun_out/r3ax.err; exit 1; }
done
DTF_DETERMINISTIC=2 timeout -k 10 300 python -u scripts/route_table.py inception_v3 256 2>>gpurun_out/r3ax.err | grep -i "MIOpen\|calls per" | cut -c1-200 | tee -a gpurun_out/r3ax.txt
for d in 1 2 1 2; do echo "bench inception_v3 det=$d $(DTF_DETERMINISTIC=$d timeout -k 10 400 python -u bench.py --model inception_v3 --steps 30 --warmup 8 2>>gpurun_out/r3ax.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")" | tee -a gpurun_out/r3ax.txt; done
