#!/usr/bin/env python3
"""K7 BatchNorm passes: [row chunk][64-channel tile] mapping vs the contiguous "sweep" mapping, on ResNet-50's
batch-256 tensors, COLD (three buffer sets rotate so that nothing is left in the 256 MiB Infinity Cache), interleaved
rounds in one process.  Forward = statistics + finalize + apply(+residual, ReLU, mask); backward = reduce + finalize +
apply (+ residual gradient)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_tensorflow_amd.ops import batchnorm as bnmod, hip

lib = hip.lib()
hip.declare("dtf_bn_set_sweep", [ctypes.c_int], None)
dev = torch.device("cuda:0")
SHAPES = [(256, 56, 256, True), (256, 56, 64, False), (256, 28, 512, True), (256, 28, 128, False), (256, 14, 1024, True),
          (256, 14, 256, False), (256, 7, 2048, True), (256, 112, 64, False)]
NSETS = 3


def timed(fn, n):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for nb, hw, c, with_res in SHAPES:
    xs = [torch.randn(nb, c, hw, hw, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
          for _ in range(NSETS)]
    rs = [torch.randn_like(x) for x in xs] if with_res else [None] * NSETS
    gs = [torch.randn_like(x) for x in xs]
    bn = bnmod.FusedBatchNormAct(c, relu=True).to(dev).to(torch.bfloat16).train()
    row = {"HW": hw, "C": c, "residual": with_res, "MB": round(xs[0].numel() * 2 / 1e6, 1)}
    res = {0: {"fwd": [], "bwd": []}, 1: {"fwd": [], "bwd": []}}
    for rnd in range(4):
        for mode in (0, 1):
            lib.dtf_bn_set_sweep(mode)

            def fwd(i):
                with torch.no_grad():
                    return bn(xs[i % NSETS], rs[i % NSETS])

            for i in range(2):
                fwd(i)
            res[mode]["fwd"].append(timed(fwd, 9))
            # backward alone: graphs built beforehand, timed region = autograd backward of the BN node only
            outs = []
            for i in range(NSETS):
                x = xs[i].detach().requires_grad_(True)
                r = rs[i].detach().requires_grad_(True) if with_res else None
                outs.append((bn(x, r), x, r))

            def bwd(i):
                y, x, r = outs[i % NSETS]
                torch.autograd.grad(y, [x] + ([r] if r is not None else []) + [bn.weight, bn.bias], gs[i % NSETS],
                                    retain_graph=True)

            for i in range(2):
                bwd(i)
            res[mode]["bwd"].append(timed(bwd, 9))
    for mode, name in ((0, "ranges"), (1, "sweep")):
        for k in ("fwd", "bwd"):
            row["%s_%s_us" % (name, k)] = round(sorted(res[mode][k])[len(res[mode][k]) // 2], 1)
    # bytes a pass sequence must move: fwd = x twice + y (+res) ; bwd = 2 x (dy + x) + dx (+dres)
    e = xs[0].numel() * 2
    row["sweep_fwd_TBps"] = round(e * (3 + (1 if with_res else 0)) / row["sweep_fwd_us"] / 1e6, 2)
    row["sweep_bwd_TBps"] = round(e * (5 + (1 if with_res else 0)) / row["sweep_bwd_us"] / 1e6, 2)
    row["ranges_fwd_TBps"] = round(e * (3 + (1 if with_res else 0)) / row["ranges_fwd_us"] / 1e6, 2)
    row["ranges_bwd_TBps"] = round(e * (5 + (1 if with_res else 0)) / row["ranges_bwd_us"] / 1e6, 2)
    print(json.dumps(row), flush=True)
    del xs, rs, gs, outs
    torch.cuda.empty_cache()
