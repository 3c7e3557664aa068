#!/bin/bash
# per-kernel times of K21 (patch matrix + K6T) at Inception-v3's stem shape, batch 256
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_k21 -o t -- python3 $R/scripts/k21_probe.py 256 2>&1 | grep -v "^[EW]2026" | tail -2
cd $R && python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/prof_k21/t_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if any(k in n for k in ("im2col", "gemm_tn", "tn_reduce")):
        print("%-60s %8.1f us  grid %s x %s" % (n[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"]))
PY
rm -rf gpurun_out/prof_k21
