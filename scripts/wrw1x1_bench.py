#!/usr/bin/env python3
"""1x1 convolution weight gradient: MIOpen (shipped tuning) vs K6T, ResNet-50 (b256) and Inception-v3 (b128) shapes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen
miopen.setup()
import torch
from distributed_tensorflow_amd.ops import hip
from distributed_tensorflow_amd.ops.linear import gemm_tn

hip.lib()
dev = torch.device("cuda:0")
# (batch, HW, Cin, Cout, count)
SHAPES = [(256, 56, 64, 64, 1), (256, 56, 64, 256, 4), (256, 56, 256, 64, 2), (256, 56, 256, 128, 1), (256, 28, 128, 512, 4),
          (256, 28, 512, 128, 3), (256, 28, 512, 256, 1), (256, 14, 256, 1024, 6), (256, 14, 1024, 256, 5),
          (128, 73, 64, 80, 1), (128, 35, 192, 64, 2), (128, 35, 256, 64, 3), (128, 35, 288, 64, 3), (128, 35, 288, 48, 1),
          (128, 17, 768, 192, 12), (128, 17, 768, 128, 2), (128, 17, 768, 160, 4), (128, 8, 1280, 320, 1), (128, 8, 1280, 384, 1),
          (128, 8, 1280, 448, 1), (128, 8, 2048, 384, 1), (128, 8, 2048, 448, 1), (128, 8, 2048, 192, 1)]


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


for nb, hw, cin, cout, cnt in SHAPES:
    x = torch.randn(nb, cin, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(nb, cout, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 1, 1, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x2 = x.permute(0, 2, 3, 1).reshape(-1, cin)
    dy2 = dy.permute(0, 2, 3, 1).reshape(-1, cout)
    r = {"b": nb, "HW": hw, "Cin": cin, "Cout": cout, "count": cnt, "M": x2.shape[0]}
    r["miopen_wrw"] = round(timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])), 1)
    if x2.shape[0] % 32 == 0 and cin % 8 == 0 and cout % 8 == 0:
        r["k6t"] = round(timeit(lambda: gemm_tn(dy2, x2)), 1)
        r["k6t_swapped"] = round(timeit(lambda: gemm_tn(x2, dy2)), 1)
    r["roofline_us_at_5TBs"] = round(x2.shape[0] * (cin + cout) * 2 / 5e12 * 1e6, 1)
    print(json.dumps(r), flush=True)
