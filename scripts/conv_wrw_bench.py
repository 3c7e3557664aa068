#!/usr/bin/env python3
"""Weight gradient of the non-1x1 convolutions: MIOpen (shipped tuning, incl. its workspace passes) vs K6T-conv."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen
miopen.setup()
import torch
from distributed_tensorflow_amd.ops import conv, hip

hip.lib()
dev = torch.device("cuda:0")
# NB, Cin, H, W, Cout, KH, KW, stride, pad, count
SHAPES = [(256, 64, 56, 56, 64, 3, 3, 1, (1, 1), 3), (256, 128, 56, 56, 128, 3, 3, 2, (1, 1), 1), (256, 128, 28, 28, 128, 3, 3, 1, (1, 1), 3),
          (256, 256, 28, 28, 256, 3, 3, 2, (1, 1), 1), (256, 256, 14, 14, 256, 3, 3, 1, (1, 1), 5), (256, 512, 14, 14, 512, 3, 3, 2, (1, 1), 1),
          (256, 512, 7, 7, 512, 3, 3, 1, (1, 1), 2),
          (256, 48, 35, 35, 64, 5, 5, 1, (2, 2), 3), (256, 64, 35, 35, 96, 3, 3, 1, (1, 1), 4), (256, 96, 35, 35, 96, 3, 3, 1, (1, 1), 3),
          (256, 288, 35, 35, 384, 3, 3, 2, (0, 0), 1), (256, 160, 17, 17, 160, 1, 7, 1, (0, 3), 6), (256, 160, 17, 17, 192, 7, 1, 1, (3, 0), 4),
          (256, 192, 17, 17, 192, 1, 7, 1, (0, 3), 6), (256, 192, 17, 17, 320, 3, 3, 2, (0, 0), 1), (256, 384, 8, 8, 384, 1, 3, 1, (0, 1), 4),
          (256, 448, 8, 8, 384, 3, 3, 1, (1, 1), 2)]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = {"miopen": 0.0, "k6t": 0.0, "best": 0.0}
for NB, Cin, H, W, Cout, KH, KW, st, pad, cnt in SHAPES:
    x = torch.randn(NB, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, KH, KW, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.nn.functional.conv2d(x, w, None, st, pad)
    dy = torch.randn_like(y).contiguous(memory_format=torch.channels_last)
    r = {"Cin": Cin, "HW": H, "Cout": Cout, "k": "%dx%d" % (KH, KW), "s": st, "count": cnt}
    r["miopen"] = round(timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], list(pad), [1, 1], False, [0, 0], 1, [False, True, False])), 1)
    r["k6t"] = round(timeit(lambda: conv.conv_wrw(dy, x, w, (st, st), pad)), 1)
    if os.environ.get("DTF_BENCH_FWD_DGRAD"):
        r["fwd"] = round(timeit(lambda: torch.nn.functional.conv2d(x, w, None, st, pad)), 1)
        r["dgrad"] = round(timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], list(pad), [1, 1], False, [0, 0], 1, [True, False, False])), 1)
    flops = 2.0 * dy.shape[0] * dy.shape[2] * dy.shape[3] * Cout * Cin * KH * KW
    r["k6t_TF"] = round(flops / r["k6t"] / 1e6, 0)
    if "fwd" in r:
        r["fwd_TF"] = round(flops / r["fwd"] / 1e6, 0)
        r["dgrad_TF"] = round(flops / r["dgrad"] / 1e6, 0)
    tot["miopen"] += r["miopen"] * cnt; tot["k6t"] += r["k6t"] * cnt; tot["best"] += min(r["miopen"], r["k6t"]) * cnt
    print(json.dumps(r), flush=True)
print(json.dumps({k: round(v, 1) for k, v in tot.items()}))
