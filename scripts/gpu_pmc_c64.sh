#!/bin/bash
# PMC counters of K16 (conv3x3_c64_kernel) and K15 (stem kernels): one counter group per run
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
rm -rf $R/gpurun_out/pmcc64; mkdir -p $R/gpurun_out/pmcc64
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcc64 -o c64_$name -- python3 $R/scripts/conv_c64_bench.py 64 2>&1 | grep -v "^[EW]2026" | tail -1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcc64 -o stem_$name -- python3 $R/scripts/stem_bench.py 64 2>&1 | grep -v "^[EW]2026" | tail -1 || exit 1; }
run mfma SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE \
 && run ldsc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE \
 && run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
cd $R && python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcc64/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'conv3x3_c64' in n or 'stem_fwd' in n or 'stem_wrw_kernel' in n:
            d[n[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_c64.txt', 'w') as out:
    for k, v in sorted(d.items()):
        out.write(k + '\n')
        for c, x in sorted(v.items()):
            out.write('    %-28s %14d\n' % (c, round(sorted(x)[len(x) // 2])))
PY
rm -rf gpurun_out/pmcc64
cat gpurun_out/pmc_c64.txt
