#!/bin/bash
# PMC counters of K13 (own convolution), one rocprofv3 pass per counter group (kernel-trace only).
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
mkdir -p $R/gpurun_out/pmcc
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcc -o $name -- python3 $R/scripts/pmc_conv.py 2>&1 | grep -v "^[EW]2026" | tail -2 || exit 1
}
run mfma SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE \
 && run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
 && run fetch FETCH_SIZE GRBM_GUI_ACTIVE
cd $R && python3 scripts/summarize_pmc.py gpurun_out/pmcc > gpurun_out/pmcc/summary.txt; cat gpurun_out/pmcc/summary.txt
