"""K7 kernel bandwidth on ResNet-50 shapes (batch 256): forward (stats+finalize+apply) and
backward (reduce+finalize+apply) time and effective HBM rate from the minimal byte counts."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import batchnorm as bn

dev = torch.device("cuda")
shapes = [(256, 64, 112, 112), (256, 64, 56, 56), (256, 256, 56, 56), (256, 128, 28, 28), (256, 512, 28, 28),
          (256, 256, 14, 14), (256, 1024, 14, 14), (256, 512, 7, 7), (256, 2048, 7, 7)]


def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for shp in shapes:
    N, C, H, W = shp
    x = torch.randn(shp, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn(shp, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(shp, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.ones(C, device=dev, dtype=torch.bfloat16, requires_grad=True)
    b = torch.zeros(C, device=dev, dtype=torch.bfloat16, requires_grad=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbytes = x.numel() * 2
    row = {"shape": shp, "MB": round(nbytes / 1e6, 1)}
    for res in (False, True):
        xr = x.clone().requires_grad_()
        rr = r.clone().requires_grad_() if res else None
        tf = timeit(lambda: bn.fused_bn_act(xr, w, b, rr, rm, rv, True, 0.1, 1e-5, True))
        y = bn.fused_bn_act(xr, w, b, rr, rm, rv, True, 0.1, 1e-5, True)
        tb = timeit(lambda: torch.autograd.grad(y, [xr, w, b] + ([rr] if res else []), dy, retain_graph=True))
        fb = nbytes * (3 + (1 if res else 0))            # stats R, apply R (+res R) W
        bb = nbytes * (2 + 3 + (1 if res else 0))        # reduce R dy,x ; apply R dy,x W dx (+W dres)
        tag = "res" if res else "plain"
        row[tag + "_fwd_us"] = round(tf * 1e6, 1); row[tag + "_fwd_TBs"] = round(fb / tf / 1e12, 2)
        row[tag + "_bwd_us"] = round(tb * 1e6, 1); row[tag + "_bwd_TBs"] = round(bb / tb / 1e12, 2)
    print(json.dumps(row), flush=True)

# calibration: what a plain streaming kernel reaches on this part (framework copy / fill / read)
for mb in (103, 411, 1644):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16).normal_()
    c = torch.empty_like(a)
    tc = timeit(lambda: c.copy_(a))
    tf = timeit(lambda: c.fill_(1.0))
    tr = timeit(lambda: a.view(torch.int16).max())
    print(json.dumps({"calib_MB": mb, "copy_TBs": round(2 * n * 2 / tc / 1e12, 2), "fill_TBs": round(n * 2 / tf / 1e12, 2),
                      "read_TBs": round(n * 2 / tr / 1e12, 2)}), flush=True)
