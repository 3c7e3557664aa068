#!/bin/bash
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
mkdir -p $R/gpurun_out/pmcw
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcw -o $name -- python3 $R/scripts/pmc_wrw.py 2>&1 | grep -v "^[EW]2026" | tail -1 || exit 1; }
run mfma SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE \
 && run fetch FETCH_SIZE GRBM_GUI_ACTIVE \
 && run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE \
 && run ldsc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE \
 && run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
cd $R && python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcw/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'gemm_tn_kernel' in n or 'igemm_wrw' in n or 'wrw_direct' in n:
            d[n[:60] + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(d.items()):
    print(k)
    print('   ', {c: round(sorted(x)[len(x) // 2]) for c, x in sorted(v.items())})
PY
