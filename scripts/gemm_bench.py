"""K6 GEMM throughput vs torch (hipBLASLt) on the framework's linear shapes.  Interleaved
rounds in one process, random data (cdna_hip_programming.md rules 24/25)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import linear as lin

dev = torch.device("cuda")
shapes = [(8192, 768, 768), (8192, 3072, 768), (8192, 768, 3072), (32768, 768, 768), (32768, 2304, 768), (32768, 3072, 768),
          (32768, 768, 3072), (65536, 1024, 1672), (65536, 512, 1024), (4096, 4096, 4096), (8192, 8192, 8192),
          (802816, 256, 64)]
if os.environ.get("DTF_GEMM_SHAPES"):          # "M,N,K;M,N,K;..."
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["DTF_GEMM_SHAPES"].split(";")]
from distributed_tensorflow_amd.ops import hip as _hip
_lib = _hip.lib()
lin._lib()
_hip.declare("dtf_gemm_set_tile", [__import__("ctypes").c_int], None)


def tiled(t, f):
    def g():
        _lib.dtf_gemm_set_tile(t)
        r = f()
        _lib.dtf_gemm_set_tile(0)
        return r
    return g
res = []
for M, N, K in shapes:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    fns = {"k6_fused_gelu": lambda: lin.gemm_bias_act(x, w, b, "gelu"),
           "k6_128": tiled(128, lambda: lin.gemm_bias_act(x, w, None, "none")),
           "k6_256": tiled(256, lambda: lin.gemm_bias_act(x, w, None, "none")),
           "torch_linear": lambda: torch.nn.functional.linear(x, w, b),
           "torch_linear_gelu": lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b))}
    times = {k: [] for k in fns}
    for rnd in range(5):
        for k, f in fns.items():
            f(); torch.cuda.synchronize()
            n = 10
            t0 = time.perf_counter()
            for _ in range(n): f()
            torch.cuda.synchronize()
            times[k].append((time.perf_counter() - t0) / n)
    row = {"M": M, "N": N, "K": K}
    if K % 64 == 0:
        ref = (x[:512].float() @ w.float().t() + b.float())
        got = tiled(256, lambda: lin.gemm_bias_act(x, w, b, "none"))()[0]
        row["err256"] = round(float((got[:512].float() - ref).abs().max()), 4)
        row["err256_tail"] = round(float((got[-300:].float() - (x[-300:].float() @ w.float().t() + b.float())).abs().max()), 4)
    for k, v in times.items():
        t = sorted(v)[len(v) // 2]
        row[k + "_us"] = round(t * 1e6, 1)
        row[k + "_TF"] = round(2.0 * M * N * K / t / 1e12, 1)
    res.append(row)
    print(json.dumps(row), flush=True)
