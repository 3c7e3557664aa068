"""K10 LayerNorm forward / backward time on BERT's [32768, 768] activations (cold: 6 rotating tensors)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import layernorm

dev = torch.device("cuda")
ln = layernorm.FusedLayerNorm(768, eps=1e-12).to(dev).to(torch.bfloat16)
xs = [torch.randn(32768, 768, device=dev).to(torch.bfloat16).requires_grad_() for _ in range(6)]
dy = torch.randn(32768, 768, device=dev).to(torch.bfloat16)
def t(fn, n=30):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
with torch.no_grad():
    f = t(lambda i: ln(xs[i % 6]))
ys = [ln(x) for x in xs]
def bwd(i):
    ys[i % 6].backward(dy, retain_graph=True)
b = t(bwd)
print("cap=%s ln_fwd %.1f us  ln_bwd(+finalize) %.1f us" % (os.environ.get("DTF_LN_FWD_GRID_CAP", "default"), f, b))
