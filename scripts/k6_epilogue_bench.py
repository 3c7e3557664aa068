#!/usr/bin/env python3
"""K6 epilogue cost: plain vs bias vs bias+residual on BERT's N = 768 shapes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
import torch.nn.functional as F
from distributed_tensorflow_amd.ops import linear as lin

dev = torch.device("cuda")


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


for M, N, K in [(32768, 768, 768), (32768, 768, 3072), (32768, 2304, 768), (32768, 3072, 768)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16)
    row = {"M": M, "N": N, "K": K}
    row["plain"] = round(timeit(lambda: lin.gemm_bias_act(x, w, None, "none")), 1)
    row["bias"] = round(timeit(lambda: lin.gemm_bias_act(x, w, b, "none")), 1)
    row["bias_res"] = round(timeit(lambda: lin.gemm_bias_act(x, w, b, "none", r)), 1)
    row["bias_gelu_tanh_preact"] = round(timeit(lambda: lin.gemm_bias_act(x, w, b, "gelu_tanh", None, save_preact=True)), 1)
    row["torch_addmm"] = round(timeit(lambda: torch.addmm(b, x, w.t())), 1)
    row["torch_addmm_residual_beta1"] = round(timeit(lambda: torch.addmm(r, x, w.t())), 1)
    y = torch.addmm(b, x, w.t())
    row["torch_gelu_pass"] = round(timeit(lambda: F.gelu(y, approximate="tanh")), 1)
    row["torch_add_pass"] = round(timeit(lambda: y + r), 1)
    print(json.dumps(row), flush=True)
