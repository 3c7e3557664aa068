#!/bin/bash
# round-3 PMC table over whole steps (ResNet-50 b256, Wide&Deep b65536): scripts/pmc_kernels3.py
set -o pipefail
PMC_TIMEOUT=300 bash scripts/gpu_pmc.sh pmc_kernels3.py pmc3 && python3 scripts/summarize_pmc_step.py gpurun_out/pmc3 > gpurun_out/pmc3_summary.txt && head -70 gpurun_out/pmc3_summary.txt
