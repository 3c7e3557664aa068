#!/bin/bash
# PMC counters of the hand-written kernels, one rocprofv3 pass per counter group (kernel-trace only).
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
DRV=${1:-pmc_kernels.py}; OUT=${2:-pmc}        # driver script under scripts/, output directory under gpurun_out/
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/$OUT -o $name -- python3 $R/scripts/$DRV 2>&1 | grep -v "^[EW]2026" | tail -2 || exit 1
}
run mfma SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
 && run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
 && run fetch FETCH_SIZE GRBM_GUI_ACTIVE \
 && run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls $R/gpurun_out/$OUT
