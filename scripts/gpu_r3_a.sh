#!/bin/bash
# round 3, call A: new multi-GPU test file (W=1 legs on this box), distributed GPU tests, bench, REST W=1 GPU job
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import torch;print('devices', torch.cuda.device_count())" > gpurun_out/r3a_devices.txt 2>&1
timeout -k 10 900 python -u -m pytest tests/test_multigpu.py tests/test_distributed_gpu.py -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/r3a_pytest_multigpu.log 2>&1
rc=$?; tail -5 gpurun_out/r3a_pytest_multigpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 8 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err && cat gpurun_out/r3a_bench.json
