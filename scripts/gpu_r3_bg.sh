#!/bin/bash
# side-stream priority experiment: does a low-priority weight-gradient stream (or a high-priority one) change the step?
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/wgrad_priority.txt
: > $O
timeout -k 10 60 python -c "
import torch
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print('asked',p,'got',s.priority)
    except Exception as e: print('asked',p,'error',e)
" 2>&1 | grep -v amdgpu.ids | tee -a $O || exit 1
for pr in 0 1 -1 0 1; do
  echo "== DTF_WGRAD_PRIORITY=$pr" | tee -a $O
  DTF_WGRAD_PRIORITY=$pr timeout -k 10 170 python bench.py --steps 60 --warmup 15 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O || exit 1
done
