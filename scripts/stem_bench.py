"""K15 stem kernels vs the library path at the bench shape (batch 256, 224 x 224): per-call times with HIP events.

    python scripts/stem_bench.py [batch]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache  # noqa: E402

miopen_cache.setup()
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from distributed_tensorflow_amd.ops import batchnorm as bnmod, pool, stem  # noqa: E402


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    x = torch.randn(N, 3, 224, 224, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    bn = bnmod.FusedBatchNormAct(64, relu=True).to(dev).to(torch.bfloat16).train()
    w = conv.weight
    print("conv fwd  library          %8.1f us" % timeit(lambda: F.conv2d(x, w, None, 2, 3)))
    print("conv fwd  K15 (+ stats)    %8.1f us" % timeit(lambda: stem.stem_conv_fwd(x, w)))
    for name, on in (("library node (conv, stats, pool | pool', bn bwd, wrw)", False), ("K15 node", True)):
        stem._ENABLED = on

        def fwd_bwd():
            y = stem.stem_conv_bn_relu_pool(x, conv, bn, 3, 2, 1)
            y.backward(y)
            conv.weight.grad = None
            bn.weight.grad = None
            bn.bias.grad = None

        print("%-55s %8.1f us fwd+bwd" % (name, timeit(fwd_bwd)))
    if hasattr(stem._lib(), "dtf_stem_wrw"):
        stem._ENABLED = True
        for on in (False, True):
            stem._WRW = on

            def fwd_bwd2():
                y = stem.stem_conv_bn_relu_pool(x, conv, bn, 3, 2, 1)
                y.backward(y)
                conv.weight.grad = None

            print("K15 node, fused weight gradient %s %8.1f us fwd+bwd" % ("on " if on else "off", timeit(fwd_bwd2)))


if __name__ == "__main__":
    main()
