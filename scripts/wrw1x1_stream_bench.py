#!/usr/bin/env python3
"""K19 (streaming 1x1 weight gradient, 64 <-> 256 channels at 56 x 56) vs MIOpen and vs K6T: numerics + time.

    python scripts/wrw1x1_stream_bench.py > gpurun_out/k19_wrw1x1.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distributed_tensorflow_amd.utils import miopen  # noqa: E402

miopen.setup()
import torch  # noqa: E402

from distributed_tensorflow_amd.ops import conv  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # evict the operands from the Infinity Cache between calls by READING 512 MB of something else (a fill would
    # leave 256 MB of dirty lines whose write-back competes with the timed kernel's reads: every kernel then
    # measures ~3.2 TB/s), so every call streams from HBM like in the model
    junk = torch.zeros(256 << 20, dtype=torch.bfloat16, device="cuda")
    ts = []
    for _ in range(n):
        junk.sum()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    N, H = 256, 56
    print("K19 streaming 1x1 weight gradient, batch %d, %dx%d (M = %d pixels)" % (N, H, H, N * H * H))
    for cout, cin in ((256, 64), (64, 256)):
        x = torch.randn(N, cin, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = (torch.randn(N, cout, H, H, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16)
        dy2, x2 = conv._rows(dy), conv._rows(x)
        assert conv.wrw_stream_ok(dy2, x2)
        got = conv.wrw_stream(dy2, x2).float()
        ref = torch.zeros(cout, cin, device=dev, dtype=torch.float32)
        for lo in range(0, dy2.shape[0], 1 << 17):            # fp32 reference in chunks
            ref += dy2[lo:lo + (1 << 17)].float().t() @ x2[lo:lo + (1 << 17)].float()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        again = conv.wrw_stream(dy2, x2).float()
        t_k19 = timeit(lambda: conv.wrw_stream(dy2, x2))
        t_mi = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                                  [False, True, False]))
        from distributed_tensorflow_amd.ops.linear import gemm_tn, gemm_tn_eligible

        t_k6t = timeit(lambda: gemm_tn(dy2, x2)) if gemm_tn_eligible(dy2, x2) else float("nan")
        mb = (dy2.numel() + x2.numel()) * 2 / 1e6
        print("  %3d -> %3d channels: max rel err vs fp32 %.2e, bitwise repeatable %s | K19 %.1f us (%.2f TB/s) | MIOpen %.1f us | "
              "K6T %.1f us" % (cin, cout, err, bool(torch.equal(got, again)), t_k19, mb / t_k19, t_mi, t_k6t))
        assert err < 2e-2


if __name__ == "__main__":
    main()
