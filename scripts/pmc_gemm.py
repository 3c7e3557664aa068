"""GEMM-only driver for rocprofv3 --pmc: K6 256-tile kernel (plain / fused) and K6T on random data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import linear as lin

dev = torch.device("cuda")
torch.manual_seed(0)
for (M, N, K) in [(4096, 4096, 4096), (32768, 3072, 768)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    for _ in range(3):
        lin.gemm_bias_act(x, w, None, "none")
    for _ in range(3):
        lin.gemm_bias_act(x, w, b, "gelu_tanh")
a = torch.randn(32768, 3072, device=dev).to(torch.bfloat16)
c = torch.randn(32768, 768, device=dev).to(torch.bfloat16)
for _ in range(3):
    lin.gemm_tn(a, c)
torch.cuda.synchronize()
print("done")
