#!/bin/bash
# kernel trace of one model; keeps the per-dispatch CSV under gpurun_out/prof_<model>/ and the step summary next to it
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; m=$1; shift
rm -rf $R/gpurun_out/prof_$m; mkdir -p $R/gpurun_out/prof_$m
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$m -o trace -- python3 $R/bench.py --model $m --steps 4 --warmup 3 "$@" 2>&1 | grep -v "^[EW]2026" | tail -1 | cut -c1-200 || exit 1
cd $R && python3 scripts/summarize_trace.py gpurun_out/prof_$m/trace_kernel_trace.csv > gpurun_out/prof_$m.txt || exit 1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/prof_$m/trace_kernel_trace.csv")))
# keep only the columns used later (the raw CSV is tens of MB)
with open("gpurun_out/prof_$m/dispatches.tsv", "w") as f:
    for r in rows[-4000:]:
        f.write("%s\t%d\t%s\t%s\n" % (r["Kernel_Name"][:90], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")))
PY
rm -f gpurun_out/prof_$m/trace_kernel_trace.csv gpurun_out/prof_$m/*agent_info.csv
sed -n 1,12p gpurun_out/prof_$m.txt
