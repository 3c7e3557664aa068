#!/bin/bash
# A/B of two builds of the kernel library in one box: libdtf_hip.so vs libdtf_hip_alt.so (GEMM table + K13 shapes)
set -o pipefail
export DTF_GEMM_SHAPES="32768,768,768;32768,2304,768;32768,3072,768;32768,768,3072;65536,512,1024;4096,4096,4096;8192,8192,8192"
for rep in ${REPS:-1 2}; do
for lib in ${LIBS:-libdtf_hip.so libdtf_hip_alt.so}; do
  echo "== $lib"
  DTF_HIP_LIB=$PWD/distributed_tensorflow_amd/ops/$lib timeout -k 10 200 python -u scripts/gemm_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['M'], r['N'], r['K'], 'k6_256', r.get('k6_256_us'), r.get('k6_256_TF'), 'lib', r.get('torch_linear_us'), r.get('torch_linear_TF'), 'err', r.get('err256'))
" || exit 1
  DTF_HIP_LIB=$PWD/distributed_tensorflow_amd/ops/$lib DTF_BENCH_SHAPES=4,6,12,15 timeout -k 10 200 python -u scripts/conv_fwd_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)
        if 'Cin' in r: print(r['Cin'], r['HW'], r['Cout'], r['k'], 'miopen', r['miopen'], 'auto', r['k13_auto'], 'best_TF', r['k13_best_TF'], 'err', r['max_rel_err'])
" || exit 1
done
done
