#!/bin/bash
# The GPU suite once against the -DDTF_DEBUG_BOUNDS build of the kernel library (index asserts by printf; no
# sanitizer, no debugger).  Build it first on the CPU box: python -m distributed_tensorflow_amd.ops.build --debug-bounds
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
export DTF_HIP_LIB=$PWD/distributed_tensorflow_amd/ops/libdtf_hip_debug.so
test -f "$DTF_HIP_LIB" || { echo "debug library missing"; exit 2; }
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -s 2>&1 > gpurun_out/debug_bounds_pytest.log
rc=$?
tail -3 gpurun_out/debug_bounds_pytest.log
n=$(grep -c DTF_BOUNDS_VIOLATION gpurun_out/debug_bounds_pytest.log)
echo "bounds violations reported: $n" | tee gpurun_out/debug_bounds_summary.txt
grep -m 5 DTF_BOUNDS_VIOLATION gpurun_out/debug_bounds_pytest.log
test $rc -eq 0 -a "$n" -eq 0
