#!/usr/bin/env python3
"""K13 statistics epilogue cost per tile variant (isolated, interleaved)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import conv, hip
hip.lib(); lib = conv._fwd_lib(); dev = torch.device("cuda:0")
def timeit(fns, n=20, rounds=3):
    res = [[] for _ in fns]
    for fn in fns: fn(); fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for i, fn in enumerate(fns):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn()
            b.record(); torch.cuda.synchronize()
            res[i].append(a.elapsed_time(b) / n * 1e3)
    return [round(sorted(r)[len(r) // 2], 1) for r in res]
for (NB, C, H, Co, st, tiles) in [(256, 512, 7, 512, 1, [(4, 4), (3, 5), (2, 8)]), (256, 512, 14, 512, 2, [(4, 4)]), (256, 256, 14, 256, 1, [(4, 7), (4, 8)])]:
    x = torch.randn(NB, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, C, 3, 3, device=dev) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    fns = []
    for t in tiles:
        for s in (False, True):
            def f(t=t, s=s):
                lib.dtf_conv_fwd_set_tile(*t); conv.conv_fwd(x, w, (st, st), (1, 1), stats=s)
            fns.append(f)
    print(C, H, st, tiles, timeit(fns), flush=True)
