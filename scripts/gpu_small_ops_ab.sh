#!/bin/bash
set -o pipefail
timeout -k 10 500 python -u -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q --timeout 300 --timeout-method thread -k "subsample or global_avg or parameter_layout or conv or resnet or inception" > gpurun_out/small_ops_tests.log 2>&1 || { tail -30 gpurun_out/small_ops_tests.log; exit 1; }
tail -2 gpurun_out/small_ops_tests.log
b() { env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$*', r['value'], r['ms_per_step'])"; }
b DTF_SUBSAMPLE_KERNEL=0 && b DTF_SUBSAMPLE_KERNEL=1 && b DTF_SUBSAMPLE_KERNEL=0 && b DTF_SUBSAMPLE_KERNEL=1
timeout -k 10 300 python -u scripts/find_small_ops.py 2>&1 | tail -14
