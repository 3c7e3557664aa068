#!/usr/bin/env python3
"""Numerics check of every shipped GEMM solution (gemm_db/gfx950_tunableop.csv): the GEMM as the models call it, with the
recorded solution active, against an fp32 reference -- on memory that was filled with NaN just before, so a solution that
reads a workspace / output it never initialised shows.  Prints one line per signature; exit 1 if any is wrong.

    python scripts/validate_gemms.py [csv]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    os.environ["DTF_GEMM_DB"] = os.path.abspath(sys.argv[1])
from distributed_tensorflow_amd.utils import gemm_tuning  # noqa: E402

db = gemm_tuning.setup()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

gemm_tuning.after_torch_import()
src = os.environ.get("DTF_GEMM_DB") or os.path.join(ROOT, "distributed_tensorflow_amd", "gemm_db", "gfx950_tunableop.csv")
rows = [ln.strip().split(",") for ln in open(src) if not ln.startswith("Validator") and ln.count(",") >= 3]
dev = torch.device("cuda:0")


def poison():
    # NaN into the allocator's free blocks: the next outputs / workspaces start as NaN
    junk = [torch.full((1 << 28,), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    del junk


def err(y, ref):
    if not torch.isfinite(y.float()).all():
        return float("inf")
    return float((y.float() - ref).norm() / ref.norm())


bad = 0
torch.manual_seed(0)
for op, sig, sol, t in rows:
    f = sig.split("_")
    kind, m, n, k = f[0], int(f[1]), int(f[2]), int(f[3])
    worst = 0.0
    for rep in range(3):
        if kind == "tn":
            x = torch.randn(n, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(m, k, device=dev) * k ** -0.5).to(torch.bfloat16)
            ref = x.float() @ w.float().t()
            poison()
            if op.startswith("GemmAndBias"):
                b = torch.randn(m, device=dev).to(torch.bfloat16)
                ref = ref + b.float()
                y = F.linear(x, w, b)
            else:
                y = torch.matmul(x, w.t())
            worst = max(worst, err(y, ref))
        elif kind == "nn":
            x = torch.randn(n, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(k, m, device=dev) * k ** -0.5).to(torch.bfloat16)
            ref = x.float() @ w.float()
            poison()
            y = torch.matmul(x, w)
            worst = max(worst, err(y, ref))
            # the in-place accumulate form of the fork (C == D, beta = 1)
            c = torch.randn(n, m, device=dev).to(torch.bfloat16)
            ref2 = ref + c.float()
            poison()
            c.addmm_(x, w)
            worst = max(worst, err(c, ref2))
        elif kind == "nt":
            a = (torch.randn(k, m, device=dev) * k ** -0.5).to(torch.bfloat16)
            b = torch.randn(k, n, device=dev).to(torch.bfloat16)
            ref = b.float().t() @ a.float()
            poison()
            y = torch.matmul(b.t(), a)
            worst = max(worst, err(y, ref))
        del ref
    ok = worst < 2e-2
    bad += not ok
    print("%-5s %-44s %-26s rel err %.2e" % ("ok" if ok else "WRONG", sig, sol, worst), flush=True)
print("%d of %d shipped solutions wrong" % (bad, len(rows)))
sys.exit(1 if bad else 0)
