#!/usr/bin/env python3
"""Forward convolution: MIOpen (shipped tuning DB) vs K13 (implicit GEMM on the K6 LDS-DMA pipeline), per layer shape of
ResNet-50 / Inception-v3 at batch 256, every K13 tile variant, interleaved rounds in one process.  Also checks K13
against an fp32 reference on every shape (max relative error over the tensor)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen
miopen.setup()
import torch
import torch.nn.functional as F
from distributed_tensorflow_amd.ops import conv, hip

hip.lib()
lib = conv._fwd_lib()
dev = torch.device("cuda:0")
# NB, Cin, H, W, Cout, KH, KW, stride, pad, count (layers of that shape: forward + dgrad-as-forward)
SHAPES = [(256, 64, 56, 56, 64, 3, 3, 1, (1, 1), 6), (256, 128, 56, 56, 128, 3, 3, 2, (1, 1), 1), (256, 128, 28, 28, 128, 3, 3, 1, (1, 1), 6),
          (256, 256, 28, 28, 256, 3, 3, 2, (1, 1), 1), (256, 256, 14, 14, 256, 3, 3, 1, (1, 1), 10), (256, 512, 14, 14, 512, 3, 3, 2, (1, 1), 1),
          (256, 512, 7, 7, 512, 3, 3, 1, (1, 1), 4),
          (256, 64, 35, 35, 96, 3, 3, 1, (1, 1), 8), (256, 96, 35, 35, 96, 3, 3, 1, (1, 1), 6),
          (256, 288, 35, 35, 384, 3, 3, 2, (0, 0), 1), (256, 160, 17, 17, 160, 1, 7, 1, (0, 3), 12), (256, 160, 17, 17, 192, 7, 1, 1, (3, 0), 8),
          (256, 192, 17, 17, 192, 1, 7, 1, (0, 3), 12), (256, 192, 17, 17, 320, 3, 3, 2, (0, 0), 1), (256, 384, 8, 8, 384, 1, 3, 1, (0, 1), 8),
          (256, 448, 8, 8, 384, 3, 3, 1, (1, 1), 4),
          (256, 32, 149, 149, 32, 3, 3, 1, (0, 0), 2), (256, 32, 147, 147, 64, 3, 3, 1, (1, 1), 2)]      # Inception stem 2a / 2b
VARIANTS = [(4, 8), (4, 7), (4, 6), (4, 5), (4, 4), (3, 8), (3, 7), (3, 6), (3, 5), (3, 4), (2, 8), (2, 4)]
if os.environ.get("DTF_BENCH_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["DTF_BENCH_SHAPES"].split(",")]


def timeit(fns, n=10, rounds=3):
    """interleaved rounds; -> per-function median of the round means [us]"""
    res = [[] for _ in fns]
    for fn in fns:
        fn(); fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for i, fn in enumerate(fns):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            res[i].append(a.elapsed_time(b) / n * 1e3)
    return [sorted(r)[len(r) // 2] for r in res]


tot = {"miopen": 0.0, "k13_best": 0.0, "best": 0.0}
for NB, Cin, H, W, Cout, KH, KW, st, pad, cnt in SHAPES:
    x = torch.randn(NB, Cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, KH, KW, device=dev) * (Cin * KH * KW) ** -0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(x.float(), w.float(), None, st, pad)
    r = {"Cin": Cin, "HW": H, "Cout": Cout, "k": "%dx%d" % (KH, KW), "s": st, "count": cnt}
    fns = [lambda: F.conv2d(x, w, None, st, pad)]
    errs = {}
    for nj, mi in VARIANTS:
        lib.dtf_conv_fwd_set_tile(nj, mi)
        y = conv.conv_fwd(x, w, (st, st), pad)
        torch.cuda.synchronize()
        errs["%d,%d" % (nj, mi)] = float((y.float() - ref).abs().max() / ref.abs().max())

        def f(nj=nj, mi=mi):
            lib.dtf_conv_fwd_set_tile(nj, mi)
            conv.conv_fwd(x, w, (st, st), pad)
        fns.append(f)

    def auto(stats):
        lib.dtf_conv_fwd_set_tile(0, 0)
        conv.conv_fwd(x, w, (st, st), pad, stats=stats)
    fns.append(lambda: auto(False))             # the library's own pick
    fns.append(lambda: auto(True))              # + BatchNorm statistics epilogue
    t = timeit(fns)
    flops = 2.0 * ref.shape[0] * ref.shape[2] * ref.shape[3] * Cout * Cin * KH * KW
    r["miopen"] = round(t[0], 1)
    r["miopen_TF"] = round(flops / t[0] / 1e6)
    for (nj, mi), tt in zip(VARIANTS, t[1:]):
        r["k13_%dx%d" % (32 * mi, 64 * nj)] = round(tt, 1)
    r["k13_auto"] = round(t[-2], 1)
    r["k13_auto_stats"] = round(t[-1], 1)
    best = min(t[1:-1])
    r["k13_best_TF"] = round(flops / best / 1e6)
    r["max_rel_err"] = round(max(errs.values()), 5)
    tot["miopen"] += t[0] * cnt; tot["k13_best"] += best * cnt; tot["best"] += min(t[0], best) * cnt
    print(json.dumps(r), flush=True)
    del ref
print(json.dumps({k: round(v, 1) for k, v in tot.items()}))
