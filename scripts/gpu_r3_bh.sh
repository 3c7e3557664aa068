#!/bin/bash
# Small batch, eager vs replayed: how busy is the device?  Kernel trace of bench.py at batch 64 both ways; per step:
# wall time (bench's own clock, un-profiled run) and the sum of kernel durations (profiled run) -> busy fraction.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out/small_batch
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$R/gpurun_out/small_batch/summary.txt
: > $O
for g in 0 1; do
  cd $R
  ms=$(timeout -k 10 200 python3 bench.py --batch 64 --steps 60 --warmup 15 --graph $g 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/small_batch -o g$g -- python3 $R/bench.py --batch 64 --steps 20 --warmup 12 --graph $g > /dev/null 2>&1 || exit 1
  cd $R
  python3 - $g $ms >> $O <<'P'
import csv, sys
g, ms = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open("gpurun_out/small_batch/g%s_kernel_trace.csv" % g)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 10 steps of the run: take the last 10/32 of the launches by count (32 steps in the run, equal launches per step)
n = len(rows)
tail = rows[-(n * 10 // 32):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail) / 10 / 1e6
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 10 / 1e6
print("batch 64, %s: %.2f ms/step un-profiled; under the profiler %.2f ms/step span, %.2f ms/step of kernel time (sum over both streams), %d launches/step -> device busy %.0f %% of the un-profiled step" % (
    "replayed hipGraph" if g == "1" else "eager", ms, span, busy, len(tail) // 10, 100 * min(1.0, busy / ms)))
P
done
cat $O
