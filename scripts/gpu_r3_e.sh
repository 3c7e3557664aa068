#!/bin/bash
# round 3, call E: full GPU suite; bench (graph auto = timed trial); pull-prefetch overlap evidence on BERT
set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3e_pytest_gpu.log 2>&1
rc=$?; tail -5 gpurun_out/r3e_pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 8 > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err && cat gpurun_out/r3e_bench.json || { tail gpurun_out/r3e_bench.err; exit 1; }
R=$PWD
cd /tmp && export TMPDIR=/tmp
for pf in 1 0; do
  mkdir -p $R/gpurun_out/prof_pf$pf
  DTF_FORCE_COMM=1 DTF_PULL_PREFETCH=$pf timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_pf$pf -o trace -- python3 $R/bench.py --model bert --steps 6 --warmup 4 --graph 0 > $R/gpurun_out/r3e_bert_pf$pf.json 2> $R/gpurun_out/r3e_bert_pf$pf.err || { tail $R/gpurun_out/r3e_bert_pf$pf.err; exit 1; }
done
cd $R
for pf in 1 0; do python3 scripts/prefetch_overlap.py gpurun_out/prof_pf$pf/trace_kernel_trace.csv "BERT-base b64, live 1-rank RCCL, DTF_PULL_PREFETCH=$pf"; done | tee gpurun_out/r3e_prefetch_overlap.txt
rm -f gpurun_out/prof_pf*/trace_kernel_trace.csv
