#!/bin/bash
# per-kernel times of the stem micro-benchmark (rocprofv3 kernel trace; medians over the timed calls)
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
rm -rf $R/gpurun_out/prof_stem; mkdir -p $R/gpurun_out/prof_stem
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_stem -o trace -- python3 $R/scripts/${PROF_SCRIPT:-stem_bench.py} "$@" 2>&1 | grep -v "^[EW]2026" | tail -8 || exit 1
cd $R && python3 - <<PY
import csv, collections, statistics
rows = list(csv.DictReader(open("gpurun_out/prof_stem/trace_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"][:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = sorted(d.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1]))
with open("gpurun_out/prof_stem.txt", "w") as f:
    for k, v in out[:30]:
        f.write("%9.1f us median  %5d calls  %s\n" % (statistics.median(v), len(v), k))
PY
rm -rf gpurun_out/prof_stem
cat gpurun_out/prof_stem.txt
