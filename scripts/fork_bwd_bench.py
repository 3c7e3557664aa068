#!/usr/bin/env python3
"""dX = dY.W + dSkip for the ResNet-50 bottleneck inputs (batch 256): library GEMM + separate add,
in-place addmm_ (beta=1 in the library GEMM), and the K6 MFMA GEMM with the residual in its epilogue."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_tensorflow_amd.ops import hip
from distributed_tensorflow_amd.ops.linear import gemm_bias_act

hip.lib()
dev = torch.device("cuda:0")
SHAPES = [(56, 64, 64, 1), (56, 256, 64, 2), (56, 256, 128, 1), (28, 512, 128, 3), (28, 512, 256, 1),
          (14, 1024, 256, 5), (14, 1024, 512, 1), (7, 2048, 512, 2)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = {"mm_add": 0.0, "addmm_inplace": 0.0, "k6_residual": 0.0}
for hw, cin, cout, cnt in SHAPES:
    M = 256 * hw * hw
    dy = torch.randn(M, cout, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, cin, device=dev) / cout ** 0.5).to(torch.bfloat16)
    ds = torch.randn(M, cin, device=dev).to(torch.bfloat16)
    r = {"HW": hw, "Cin": cin, "Cout": cout, "count": cnt}
    r["mm_add"] = timeit(lambda: torch.matmul(dy, w) + ds)
    buf = ds.clone()
    r["addmm_inplace"] = timeit(lambda: buf.addmm_(dy, w))
    r["k6_residual"] = timeit(lambda: gemm_bias_act(dy, w.t().contiguous(), None, "none", ds))
    ref = torch.matmul(dy.float()[:4096], w.float()) + ds.float()[:4096]
    got = gemm_bias_act(dy, w.t().contiguous(), None, "none", ds)[0][:4096].float()
    r["k6_max_err"] = float((got - ref).abs().max())
    for k in tot:
        tot[k] += r[k] * cnt
        r[k] = round(r[k], 1)
    print(json.dumps(r), flush=True)
print(json.dumps({"total_us_per_step": {k: round(v, 1) for k, v in tot.items()}}))
