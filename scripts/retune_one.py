#!/usr/bin/env python3
"""Tune ONE GEMM signature with TunableOp's numerical check on, on NaN-poisoned memory; prints the csv line to add to
gemm_db/gfx950_tunableop.csv (validate the result with scripts/validate_gemms.py before shipping it).

    python scripts/retune_one.py tn 256 802816 64
"""
import os
import sys

os.environ["DTF_GEMM_TUNING"] = "0"
import torch
import torch.cuda.tunable as tunable

kind, m, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
out = os.path.join("gpurun_out", "tunableop", "one.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
for f in os.listdir(os.path.dirname(out)):
    if f.startswith("one"):
        os.remove(os.path.join(os.path.dirname(out), f))
dev = torch.device("cuda:0")
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(out)
tunable.set_numerical_check_tolerances(True, 1e-2, 1e-2)
tunable.set_max_tuning_duration(60)
tunable.set_max_tuning_iterations(200)
torch.manual_seed(0)
junk = [torch.full((1 << 28,), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(3)]
torch.cuda.synchronize()
del junk
if kind == "tn":
    x = torch.randn(n, k, device=dev).to(torch.bfloat16)
    w = torch.randn(m, k, device=dev).to(torch.bfloat16)
    torch.matmul(x, w.t())
elif kind == "nn":
    x = torch.randn(n, k, device=dev).to(torch.bfloat16)
    w = torch.randn(k, m, device=dev).to(torch.bfloat16)
    torch.matmul(x, w)
torch.cuda.synchronize()
tunable.write_file() if hasattr(tunable, "write_file") else None
for f in sorted(os.listdir(os.path.dirname(out))):
    if f.startswith("one"):
        for ln in open(os.path.join(os.path.dirname(out), f)):
            if not ln.startswith("Validator"):
                print(ln.strip())
