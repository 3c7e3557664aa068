"""K6T on the 1x1 weight-gradient shapes of ResNet-50 (batch 256) with their HBM floor; kernel times by HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import linear as lin

dev = torch.device("cuda")
# (pixels, Cout, Cin, calls per step)
shapes = [(200704, 128, 512, 3), (200704, 512, 128, 4), (200704, 256, 512, 1), (200704, 1024, 512, 1),
          (50176, 256, 1024, 5), (50176, 1024, 256, 6), (50176, 512, 1024, 1), (50176, 2048, 1024, 1),
          (12544, 512, 2048, 2), (12544, 2048, 512, 3)]
tot = totf = 0.0
for M, N, K, calls in shapes:
    a = torch.randn(M, N, device=dev).to(torch.bfloat16)
    b = torch.randn(M, K, device=dev).to(torch.bfloat16)
    for _ in range(3):
        lin.gemm_tn(a, b)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    for s, e in ev:
        s.record(); lin.gemm_tn(a, b); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) * 1e3 for s, e in ev)[7]
    byts = M * (N + K) * 2 + N * K * 2
    floor = byts / 5.5e12 * 1e6
    tot += t * calls; totf += floor * calls
    print("M %7d N %5d K %5d x%d: %7.1f us  %5.2f TB/s  %6.0f TFLOP/s  floor(5.5 TB/s) %6.1f us" % (
        M, N, K, calls, t, byts / t / 1e6, 2.0 * M * N * K / t / 1e6, floor), flush=True)
print("per step: %.0f us, floor %.0f us" % (tot, totf))
