#!/bin/bash
# round 3, call C: submit -> step 1 with cold / warm / GPU-hot / GPU-hot + model hint ranks; prefetch test
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -u -m pytest tests/test_distributed_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread -k "prefetch or wide_deep" > gpurun_out/r3c_pytest.log 2>&1 || { tail -30 gpurun_out/r3c_pytest.log; exit 1; }
tail -2 gpurun_out/r3c_pytest.log
for cfg in "cold --warm 0 --hot 0" "warm --warm 2 --hot 0" "hot --warm 0 --hot 1" "hothint --warm 0 --hot 1 --warm-model resnet50?batch=256"; do
  set -- $cfg; tag=$1; shift
  timeout -k 10 400 python scripts/submit_latency.py --model resnet50 --steps 60 --batch 256 "$@" --out gpurun_out/r3c_submit_$tag.json > gpurun_out/r3c_submit_$tag.log 2>&1 || { tail -30 gpurun_out/r3c_submit_$tag.log; exit 1; }
  grep -E "first step|GPU-hot ranks ready" gpurun_out/r3c_submit_$tag.log
done
