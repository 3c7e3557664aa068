#!/bin/bash
# MIOpen exhaustive find for the bench's conv shapes; results (user find-db/perf-db + kernel cache)
# are written under gpurun_out/miopen so they can be shipped in-tree.
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/miopen/db gpurun_out/miopen/cache
# start from the shipped results so that new shapes are appended to them
cp -n distributed_tensorflow_amd/miopen_db/*.txt gpurun_out/miopen/db/ 2>/dev/null
cp -n distributed_tensorflow_amd/miopen_db/*.ukdb gpurun_out/miopen/cache/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen/db
export MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen/cache
( while true; do sleep 45; echo "[tune] alive $(date +%T) db=$(du -sk gpurun_out/miopen | cut -f1)KB"; done ) &
HB=$!
timeout -k 10 1000 python -u bench.py --steps 20 --warmup 4 --miopen-find 1 "$@" 2>&1 | tee gpurun_out/bench_find.log | tail -8
rc=$?
kill $HB
ls -la gpurun_out/miopen/db gpurun_out/miopen/cache | head -20
du -sh gpurun_out/miopen
exit $rc
