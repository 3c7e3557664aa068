#!/usr/bin/env python3
"""Every convolution layer shape of ResNet-50 at the bench batch size through the SAME routing as the model (Conv1x1 /
Conv2dTN modules: own kernels, tuned MIOpen solvers, recorded GEMM solutions), forward + both gradients, against fp32
conv2d / autograd on NaN-poisoned memory.  One line per layer shape; exit 1 if any relative error exceeds 2e-2.

    python scripts/validate_convs.py [batch] [model]      # model: resnet50 (fixed table) or any bench model (shapes recorded
                                                           # from one forward pass, e.g. inception_v3)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distributed_tensorflow_amd.utils import miopen as miopen_cache  # noqa: E402

miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning  # noqa: E402

gemm_tuning.setup()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import resnet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
# (cin, cout, k, stride, input H) of ResNet-50 v1.5 at 224 x 224
shapes = [(64, 64, 1, 1, 56), (64, 64, 3, 1, 56), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56), (256, 128, 1, 1, 56),
          (128, 128, 3, 2, 56), (256, 512, 1, 2, 56), (128, 512, 1, 1, 28), (512, 128, 1, 1, 28), (128, 128, 3, 1, 28),
          (512, 256, 1, 1, 28), (256, 256, 3, 2, 28), (512, 1024, 1, 2, 28), (256, 1024, 1, 1, 14), (1024, 256, 1, 1, 14),
          (256, 256, 3, 1, 14), (1024, 512, 1, 1, 14), (512, 512, 3, 2, 14), (1024, 2048, 1, 2, 14), (512, 2048, 1, 1, 7),
          (2048, 512, 1, 1, 7), (512, 512, 3, 1, 7)]


MODEL = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
layers = None
if MODEL != "resnet50":
    # record (module type, geometry, input size) of every convolution module during one forward pass
    import torch.nn as nn
    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.ops.conv import Conv1x1

    mod = get_model_module(MODEL)
    net = mod.prepare(mod.build(), dev).train()
    seen = {}

    def hook(m, inp, out):
        x = inp[0]
        key = (type(m).__name__, m.in_channels, m.out_channels, tuple(m.kernel_size), tuple(m.stride), tuple(m.padding),
               x.shape[2], x.shape[3])
        seen.setdefault(key, type(m))

    hs = [m.register_forward_hook(hook) for m in net.modules() if isinstance(m, nn.Conv2d)]
    with torch.no_grad():
        mod.loss_fn(net, mod.synthetic_batch(2, device=dev, seed=1))
    for h in hs:
        h.remove()
    del net
    layers = list(seen.items())


def poison():
    junk = [torch.full((1 << 28,), float("nan"), dtype=torch.bfloat16, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    del junk


def rel(a, b):
    a = a.float()
    if not torch.isfinite(a).all():
        return float("inf")
    return float((a - b).norm() / b.norm())


bad = 0
torch.manual_seed(0)
if layers is None:
    layers = [((None, cin, cout, (k, k), (s, s), (k // 2, k // 2), H, H), None) for cin, cout, k, s, H in shapes]
for (tname, cin, cout, ks, st, pd, H, W), cls in layers:
    k, s = ks[0], st[0]
    if cls is None:
        m = resnet._conv(cin, cout, k, s)
    elif tname == "Conv1x1":
        m = cls(cin, cout)
    else:
        m = cls(cin, cout, ks, stride=st, padding=pd, bias=False)
    m = m.to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        m.weight.normal_(0, (cin * ks[0] * ks[1]) ** -0.5)
    x = torch.randn(B, cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    xr = x.detach().float().requires_grad_()
    wr = m.weight.detach().float().requires_grad_()
    yr = F.conv2d(xr, wr, None, st, pd)
    g = torch.randn_like(yr).to(torch.bfloat16)
    yr.backward(g.float())
    poison()
    y = m(x)
    poison()
    y.backward(g.contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    e = (rel(y, yr.detach()), rel(x.grad, xr.grad), rel(m.weight.grad, wr.grad))
    ok = max(e) < 2e-2
    bad += not ok
    print("%-5s %4d -> %4d  %dx%d / %d pad %d,%d  %3d x %-3d  y %.2e  dx %.2e  dw %.2e  (%s)" % (
        "ok" if ok else "WRONG", cin, cout, ks[0], ks[1], s, pd[0], pd[1], H, W, e[0], e[1], e[2], type(m).__name__), flush=True)
    del x, xr, wr, yr, y, g, m
print("%d of %d layer shapes wrong" % (bad, len(layers)))
sys.exit(1 if bad else 0)
