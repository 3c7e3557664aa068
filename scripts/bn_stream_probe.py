#!/usr/bin/env python3
"""Why is a cold BatchNorm apply (x + residual -> y) at 3.5 TB/s when a copy does 5.3?  Apply-only passes on
[802816 x 256] bf16 (411 MB per tensor), cold (rotating buffer sets), with 1, 2, 3 streams and with the operands'
base addresses staggered."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import batchnorm as bnmod, hip

hip.lib()
dev = torch.device("cuda:0")
N, C, HW = 256, 256, 56
NSETS = 3


def timed(fn, n=9):
    for i in range(2):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def alloc(pad_bytes):
    """A channels_last [N, C, HW, HW] bf16 tensor whose storage starts `pad_bytes` into its allocation."""
    n = N * C * HW * HW
    flat = torch.empty(n + pad_bytes // 2, dtype=torch.bfloat16, device=dev).normal_()
    return flat[pad_bytes // 2:].view(N, HW, HW, C).permute(0, 3, 1, 2)


w = torch.ones(C, device=dev, dtype=torch.bfloat16)
b = torch.zeros(C, device=dev, dtype=torch.bfloat16)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
e = N * C * HW * HW * 2
for pads in ((0, 0), (0, 4096), (0, 1 << 20), (0, 3 << 20), (256, 8192 + 512)):
    xs = [alloc(pads[0]) for _ in range(NSETS)]
    rs = [alloc(pads[1]) for _ in range(NSETS)]
    row = {"pad_x": pads[0], "pad_res": pads[1]}
    with torch.no_grad():
        t = timed(lambda i: bnmod.fused_bn_act(xs[i % NSETS], w, b, None, rm, rv, False, 0.1, 1e-5, relu=False))
        row["x->y"] = [round(t, 1), round(2 * e / t / 1e6, 2)]
        t = timed(lambda i: bnmod.fused_bn_act(xs[i % NSETS], w, b, rs[i % NSETS], rm, rv, False, 0.1, 1e-5, relu=False))
        row["x+res->y"] = [round(t, 1), round(3 * e / t / 1e6, 2)]
        t = timed(lambda i: bnmod.fused_bn_act(xs[i % NSETS], w, b, rs[i % NSETS], rm, rv, False, 0.1, 1e-5, relu=True))
        row["x+res->relu y+mask"] = [round(t, 1), round(3 * e / t / 1e6, 2)]
        t = timed(lambda i: xs[i % NSETS].add(rs[i % NSETS]))
        row["torch x+res"] = [round(t, 1), round(3 * e / t / 1e6, 2)]
        t = timed(lambda i: xs[i % NSETS].clone())
        row["torch copy"] = [round(t, 1), round(2 * e / t / 1e6, 2)]
    print(json.dumps(row), flush=True)
    del xs, rs
    torch.cuda.empty_cache()
