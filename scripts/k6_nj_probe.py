#!/usr/bin/env python3
"""K6 256-row tile: 256 x 256 (NJ = 4) vs 256 x 192 (NJ = 3) on the BERT shapes, interleaved."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import linear as lin, hip
lib = hip.lib(); lin._lib()
hip.declare("dtf_gemm_set_tile", [ctypes.c_int], None); hip.declare("dtf_gemm_set_nj", [ctypes.c_int], None)
dev = torch.device("cuda")
def timeit(fns, n=10, rounds=5):
    res = [[] for _ in fns]
    for f in fns: f(); f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for i, f in enumerate(fns):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): f()
            b.record(); torch.cuda.synchronize()
            res[i].append(a.elapsed_time(b) / n * 1e3)
    return [round(sorted(r)[len(r) // 2], 1) for r in res]
for M, N, K in [(32768, 768, 768), (32768, 768, 3072), (32768, 2304, 768), (32768, 3072, 768), (8192, 768, 3072), (8192, 3072, 768)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    def mk(tile, nj, bias=None, act="none", pre=False):
        def f():
            lib.dtf_gemm_set_tile(tile); lib.dtf_gemm_set_nj(nj)
            lin.gemm_bias_act(x, w, bias, act, None, save_preact=pre)
            lib.dtf_gemm_set_tile(0); lib.dtf_gemm_set_nj(0)
        return f
    t = timeit([mk(256, 4), mk(256, 3), mk(0, 0), mk(128, 0), mk(256, 4, b, "gelu_tanh", True), mk(256, 3, b, "gelu_tanh", True), lambda: torch.addmm(b, x, w.t())])
    print(M, N, K, dict(zip(["nj4", "nj3", "auto", "t128", "nj4_gelu_pre", "nj3_gelu_pre", "lib_addmm"], t)), flush=True)
