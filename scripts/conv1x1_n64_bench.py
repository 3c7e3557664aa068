"""K18 (csrc/conv1x1_n64_kernels.hip) vs the library GEMM on the stage-1 shapes of ResNet-50 at batch 256."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import conv

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 802816


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for k in (256, 64):
    x2 = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w2 = (torch.randn(64, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    wt = w2.t().contiguous()
    us = t(lambda: conv.mm_n64(x2, w2))
    lib_nt = t(lambda: torch.matmul(x2, w2.t()))
    lib_nn = t(lambda: torch.matmul(x2, wt))
    gb = (M * k + M * 64) * 2 / 1e9
    print("K=%3d M=%d: K18 %.1f us (%.2f TB/s)   library x.w^T %.1f us   x.w %.1f us" % (k, M, us, gb / us * 1e3, lib_nt, lib_nn), flush=True)
