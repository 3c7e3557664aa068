#!/bin/bash
# What does a shape that is NOT in the shipped MIOpen kernel cache cost at start-up, and which solver families are
# responsible?  Runs bench.py at batch sizes the cache was not tuned for, with empty user caches, under several
# solver-family restrictions; prints time to the first three warm-up steps and the steady-state rate.
set -u
out=gpurun_out/miopen_miss; mkdir -p $out
run() {  # tag model batch env...
  tag=$1; model=$2; batch=$3; shift 3
  d=$(mktemp -d /tmp/miopen_probe.XXXX)
  ( if [ "${SHIPPED:-0}" = 0 ]; then export MIOPEN_USER_DB_PATH=$d/db MIOPEN_CUSTOM_CACHE_DIR=$d/cache; mkdir -p $d/db $d/cache; fi
    for kv in "$@"; do export "$kv"; done
    s=$(date +%s.%N)
    timeout -k 10 400 python -u bench.py --model $model --batch $batch --steps 10 --warmup 4 >$out/$tag.json 2>$out/$tag.err
    rc=$?
    e=$(date +%s.%N)
    python - <<PY >$out/$tag.kernels.txt 2>&1
import glob, sqlite3
for f in glob.glob("$d/cache/**/*.ukdb", recursive=True) + glob.glob("$d/cache/*.ukdb"):
    try:
        for name, args in sqlite3.connect(f).execute("select kernel_name, kernel_args from kern_db"):
            print(name, "|", args[:160])
    except Exception as ex:
        print("ukdb unreadable", f, ex)
PY
    echo "$tag rc=$rc jit_kernels=$(grep -c '|' $out/$tag.kernels.txt) $(grep -h 'warmup step' $out/$tag.err | tr '\n' ' ') $(python - <<PY
import json
try:
    j = json.loads(open("$out/$tag.json").read().strip().splitlines()[-1]); print("value=%.0f ms=%.2f" % (j["value"], j["ms_per_step"]))
except Exception as ex: print("nojson", ex)
PY
)" | tee -a $out/summary.txt
    exit $rc )
}
ONLY_IGEMM="MIOPEN_DEBUG_CONV_DIRECT=0 MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_FFT=0 MIOPEN_DEBUG_CONV_GEMM=0"
: > $out/summary.txt
case "${1:-miss}" in
miss)
run r50_b64_default resnet50 64 && \
run r50_b64_igemm resnet50 64 $ONLY_IGEMM && \
run inc_b64_default inception_v3 64 && \
run inc_b64_igemm inception_v3 64 $ONLY_IGEMM ;;
tuned)
SHIPPED=1 run r50_b256_shipped resnet50 256 && \
SHIPPED=1 run r50_b256_shipped_igemm resnet50 256 $ONLY_IGEMM && \
run r50_b256_empty_igemm resnet50 256 $ONLY_IGEMM && \
SHIPPED=1 run inc_b256_shipped inception_v3 256 && \
SHIPPED=1 run inc_b256_shipped_igemm inception_v3 256 $ONLY_IGEMM && \
run inc_b256_empty_igemm inception_v3 256 $ONLY_IGEMM ;;
esac
