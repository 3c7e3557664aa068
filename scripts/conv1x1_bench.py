"""ResNet-50 1x1 stride-1 convolutions (batch 256, NHWC bf16): MIOpen conv (tuned DB) vs the same
products as plain GEMMs through hipBLASLt (torch.matmul) -- forward, data-gradient, weight-gradient."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen
miopen.setup()
import torch
import torch.nn.functional as F

dev = torch.device("cuda")
# (HW, Cin, Cout, count in ResNet-50 v1.5)
shapes = [(56, 64, 64, 1), (56, 64, 256, 4), (56, 256, 64, 2), (56, 256, 128, 1), (28, 128, 512, 4), (28, 512, 128, 3),
          (28, 512, 256, 1), (14, 256, 1024, 6), (14, 1024, 256, 5), (14, 1024, 512, 1), (7, 512, 2048, 3), (7, 2048, 512, 2)]


def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


tot = {k: 0.0 for k in ("conv_fwd", "mm_fwd", "conv_bwd_data", "mm_bwd_data", "conv_wrw", "mm_wrw")}
for hw, ci, co, cnt in shapes:
    N = 256
    x = torch.randn(N, ci, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 1, 1, device=dev) / ci ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, co, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    M = N * hw * hw
    x2 = x.permute(0, 2, 3, 1).reshape(M, ci)
    dy2 = dy.permute(0, 2, 3, 1).reshape(M, co)
    w2 = w.reshape(co, ci)
    assert x2.is_contiguous() and dy2.is_contiguous()
    r = {"HW": hw, "Cin": ci, "Cout": co, "count": cnt}
    r["conv_fwd"] = timeit(lambda: F.conv2d(x, w))
    r["mm_fwd"] = timeit(lambda: torch.matmul(x2, w2.t()))
    r["conv_bwd_data"] = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]))
    r["mm_bwd_data"] = timeit(lambda: torch.matmul(dy2, w2))
    r["conv_wrw"] = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    r["mm_wrw"] = timeit(lambda: torch.matmul(dy2.t(), x2))
    for k in tot:
        tot[k] += r[k] * cnt
        r[k] = round(r[k], 1)
    print(json.dumps(r), flush=True)
print(json.dumps({"total_us_per_step": {k: round(v, 1) for k, v in tot.items()}}))
