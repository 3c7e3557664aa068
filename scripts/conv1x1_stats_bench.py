#!/usr/bin/env python3
"""1x1 convolution + training BatchNorm forward: library conv then K7 (stats + apply) vs K6 with the statistics
in its epilogue then K7 (finalize + apply).  ResNet-50 batch-256 expand / shortcut shapes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen, gemm_tuning
miopen.setup()
gemm_tuning.setup()
os.environ.setdefault("DTF_CONV1X1_STATS", "1")
import torch
from distributed_tensorflow_amd.ops import batchnorm as bnmod, conv, hip

hip.lib()
dev = torch.device("cuda:0")
SHAPES = [(256, 56, 64, 256, 4), (256, 28, 128, 512, 4), (256, 28, 256, 512, 0), (256, 14, 256, 1024, 6), (256, 7, 512, 2048, 3),
          (256, 35, 192, 64, 0), (256, 17, 768, 192, 0), (256, 8, 1280, 320, 0)]
if os.environ.get("DTF_BENCH_ALL_RESNET") == "1":       # every 1x1 forward product of ResNet-50 at batch 256 with Cout >= 128
    SHAPES = [(256, 56, 256, 128, 1), (256, 28, 128, 512, 4), (256, 28, 512, 128, 3), (256, 28, 256, 512, 1), (256, 28, 512, 256, 1),
              (256, 14, 256, 1024, 6), (256, 14, 1024, 256, 5), (256, 14, 512, 1024, 1), (256, 14, 1024, 512, 1),
              (256, 7, 512, 2048, 3), (256, 7, 2048, 512, 2), (256, 7, 1024, 2048, 1)]


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
    w = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn(nb, cout, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bn = bnmod.FusedBatchNormAct(cout, relu=True).to(dev).to(torch.bfloat16).train()
    r = {"HW": hw, "Cin": cin, "Cout": cout, "count": cnt}
    with torch.no_grad():
        conv._STATS_MIN_COUT = 8
        r["lib_conv"] = round(timeit(lambda: conv.conv1x1(x, w)), 1)
        r["k6_stats_conv"] = round(timeit(lambda: conv.conv1x1_stats(x, w)), 1)
        y = conv.conv1x1(x, w)
        y2, st = conv.conv1x1_stats(x, w)
        r["bn_full"] = round(timeit(lambda: bn(y, res)), 1)
        r["bn_pre"] = round(timeit(lambda: bn(y2, res, stats=st)), 1)
    r["lib_total"] = round(r["lib_conv"] + r["bn_full"], 1)
    r["k6_total"] = round(r["k6_stats_conv"] + r["bn_pre"], 1)
    print(json.dumps(r), flush=True)
