#!/bin/bash
set -o pipefail
bash scripts/gpu_profile.sh || exit 1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/prof/trace_kernel_trace.csv')))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'gemm256p_kernel' in n or 'kernel_grouped_conv_fwd' in n or n.startswith('igemm_fwd'):
        agg[n[:90] + ' grid=' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print('%6d calls  median %7.1f us  min %7.1f  %s' % (len(v), v[len(v) // 2], v[0], k))
PY
