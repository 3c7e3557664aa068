"""K6T (dW = dZ^T X) vs the library GEMM on the BERT / ResNet weight-gradient shapes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import linear as lin

dev = torch.device("cuda")
shapes = [(32768, 768, 768), (32768, 2304, 768), (32768, 3072, 768), (32768, 768, 3072), (8192, 3072, 768),
          (802816, 64, 256), (802816, 256, 64), (200704, 512, 128), (50176, 1024, 256), (12544, 2048, 512)]
for M, N, K in shapes:
    a = torch.randn(M, N, device=dev).to(torch.bfloat16)
    b = torch.randn(M, K, device=dev).to(torch.bfloat16)
    fns = {"k6t": lambda: lin.gemm_tn(a, b), "torch": lambda: torch.matmul(a.t(), b)}
    times = {k: [] for k in fns}
    for rnd in range(3):
        for k, f in fns.items():
            f(); torch.cuda.synchronize()
            n = 5
            t0 = time.perf_counter()
            for _ in range(n): f()
            torch.cuda.synchronize()
            times[k].append((time.perf_counter() - t0) / n)
    row = {"M": M, "N": N, "K": K}
    for k, v in times.items():
        t = sorted(v)[len(v) // 2]
        row[k + "_us"] = round(t * 1e6, 1)
        row[k + "_TF"] = round(2.0 * M * N * K / t / 1e12, 1)
    row["GBs_k6t"] = round((M * (N + K) * 2) / (row["k6t_us"] * 1e-6) / 1e9, 0)
    print(json.dumps(row), flush=True)
