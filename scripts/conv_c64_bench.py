"""K16 (direct 64 -> 64 channel 3x3 convolution) vs MIOpen at the ResNet-50 stage-1 shape; HIP-event medians.

    python scripts/conv_c64_bench.py [batch] [H] [W]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache  # noqa: E402

miopen_cache.setup()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from distributed_tensorflow_amd.ops import conv  # noqa: E402


def timeit(fn, n=30, warm=5):
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
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 56
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 56
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    x = torch.randn(N, 64, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device=dev) * 576 ** -0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * N * H * W * 64 * 576
    for name, fn in (("MIOpen", lambda: F.conv2d(x, w, None, 1, 1)),
                     ("K16", lambda: conv.conv_fwd(x, w, (1, 1), (1, 1))),
                     ("K16 + statistics", lambda: conv.conv_fwd(x, w, (1, 1), (1, 1), stats=True))):
        t = timeit(fn)
        print("%-18s %8.1f us  %6.0f TFLOP/s" % (name, t, fl / t / 1e6))


if __name__ == "__main__":
    main()
