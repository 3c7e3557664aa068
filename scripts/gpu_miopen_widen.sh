#!/bin/bash
# Widen the shipped MIOpen kernel cache beyond the batch-256 shapes it was tuned on.
# With the solver families restricted to implicit GEMM (utils/miopen.py does that for every task) the only kernels
# MIOpen still builds at run time are its shape-independent gfx950 assembly igemm kernels and a few tensor-op helpers;
# a sweep over models x batch sizes collects them into ONE kernel cache, which then serves shapes nobody tuned for.
# Output: gpurun_out/miopen_widen/{gfx950100.ukdb, kernels.txt, summary.txt}; copy the .ukdb over
# distributed_tensorflow_amd/miopen_db/gfx950100.ukdb.
set -u
out=gpurun_out/miopen_widen; mkdir -p $out
d=$(mktemp -d /tmp/miopen_widen.XXXX); mkdir -p $d/db $d/cache
cp distributed_tensorflow_amd/miopen_db/*.txt $d/db/
cp distributed_tensorflow_amd/miopen_db/*.ukdb $d/cache/
export MIOPEN_USER_DB_PATH=$d/db MIOPEN_CUSTOM_CACHE_DIR=$d/cache
: > $out/summary.txt
run() {  # model batch [extra bench args]
  model=$1; batch=$2; shift 2
  s=$(date +%s)
  timeout -k 10 300 python -u bench.py --model $model --batch $batch --steps 2 --warmup 2 "$@" >$out/last.json 2>$out/last.err
  rc=$?
  echo "$model b$batch $* rc=$rc $(( $(date +%s) - s )) s  ukdb=$(stat -c %s $d/cache/*.ukdb | head -1) B" | tee -a $out/summary.txt
  return $rc
}
run resnet50 128 && run resnet50 64 && run resnet50 32 && run resnet50 16 && run resnet50 512 && \
run resnet50 96 && run resnet50 64 --image-size 64 && run resnet50 8 --image-size 64 && run resnet50 4 --image-size 64 && \
run resnet50 64 --model-arg depth=101 && run resnet50 64 --model-arg depth=152 && \
run inception_v3 128 && run inception_v3 64 && run inception_v3 32 && run inception_v3 16 && run inception_v3 96
rc=$?
cp $d/cache/*.ukdb $out/ && cp $d/db/*.txt $out/
python - <<PY >$out/kernels.txt
import glob, sqlite3
for f in glob.glob("$out/*.ukdb"):
    for name, args in sqlite3.connect(f).execute("select kernel_name, kernel_args from kern_db"):
        print(name, "|", args[:120])
PY
echo "kernels: $(wc -l < $out/kernels.txt)" | tee -a $out/summary.txt
exit $rc
