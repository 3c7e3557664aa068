#!/usr/bin/env python3
"""64 -> 64 channel 3x3 weight gradient at 56x56, batch 256: MIOpen vs K6T-conv vs K14 (direct), interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen
miopen.setup()
import torch
from distributed_tensorflow_amd.ops import conv, hip
hip.lib(); dev = torch.device("cuda:0")
def timeit(fns, n=10, rounds=3):
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
for NB, H, C in [(256, 56, 64), (256, 35, 96), (256, 28, 128)]:
    x = torch.randn(NB, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(NB, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def k(direct):
        def f():
            conv._CONV_WRW_DIRECT = direct
            conv.conv_wrw(dy, x, w, (1, 1), (1, 1))
        return f
    t = timeit([lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]), k(False), k(True)])
    fl = 2.0 * NB * H * H * C * C * 9
    print(NB, H, C, dict(zip(["miopen", "k6t", "k14_direct"], t)), "k14 TF", round(fl / t[2] / 1e6), flush=True)
