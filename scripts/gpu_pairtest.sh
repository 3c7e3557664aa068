export DTF_GEMM_SHAPES="8192,8192,8192;4096,4096,8192;16384,4096,6144;4096,4096,4096;32768,3072,768;32768,768,3072;32768,2304,768"
for cfg in "1 4096" "4 4096" "4 100000" "8 100000" "2 100000" "16 100000"; do set -- $cfg; echo "gemm group_m=$1 pair_min_k=$2"; DTF_GEMM_GROUP_M=$1 DTF_GEMM_PAIR_MIN_K=$2 timeout -k 10 200 python -u scripts/gemm_bench.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['M'], r['N'], r['K'], 'k6_256', r.get('k6_256_us'), r.get('k6_256_TF'), 'lib', r.get('torch_linear_us'), r.get('torch_linear_TF'), 'err', r.get('err256'), r.get('err256_tail'))
"; done
