#!/bin/bash
# strict deterministic mode at batch sizes outside the tuned ones: products that would get MIOpen's heuristic pick stay on own kernels
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r3aw.txt
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "strict_mode or image_input_conv" 2>&1 | tail -15 || exit 1
for b in 32 48 40; do for d in 1 2; do
  echo "== resnet50 batch $b DTF_DETERMINISTIC=$d" | tee -a gpurun_out/r3aw.txt
  DTF_DETERMINISTIC=$d timeout -k 10 300 python -u scripts/determinism_check.py $b 2>gpurun_out/r3aw.err | tail -2 | cut -c1-300 | tee -a gpurun_out/r3aw.txt || { tail gpurun_out/r3aw.err; exit 1; }
done; done
DTF_DETERMINISTIC=2 timeout -k 10 200 python -u scripts/route_table.py resnet50 32 2>>gpurun_out/r3aw.err | grep -i "MIOpen\|calls per" | tee -a gpurun_out/r3aw.txt
for d in 1 2 1 2; do echo "bench det=$d $(DTF_DETERMINISTIC=$d timeout -k 10 400 python -u bench.py --steps 30 --warmup 8 2>>gpurun_out/r3aw.err | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['final_loss'])")" | tee -a gpurun_out/r3aw.txt; done
