#!/bin/bash
set -o pipefail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
mkdir -p $R/gpurun_out/pmcg
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcg -o $name -- python3 $R/scripts/pmc_gemm.py 2>&1 | grep -v "^[EW]2026" | tail -2 || exit 1
}
run lds2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM \
 && run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum TA_TA_BUSY_sum TA_BUSY_avr \
 && run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
