"""Is the FORWARD reproducible?  The loss of one fixed batch, N times (no backward), printed with full precision; plus
the first layer outputs that differ between two passes (hooks on every leaf module)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import get_model_module

if os.environ.get("DTF_CUDNN_DET") == "1":
    torch.backends.cudnn.deterministic = True
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mod = get_model_module(sys.argv[2] if len(sys.argv) > 2 else "resnet50")
m = mod.prepare(mod.build(), dev).train()
batch = mod.synthetic_batch(B, device=dev, seed=1)
outs = []
def hook(name):
    def f(mm, a, out):
        if torch.is_tensor(out):
            cur[name] = out.detach().clone()
    return f
for n, mm in m.named_modules():
    mm.register_forward_hook(hook(n))
losses = []
prev = None
first_bad = {}
for it in range(6):
    cur = {}
    with torch.enable_grad():
        loss = mod.loss_fn(m, batch)
    torch.cuda.synchronize()
    losses.append(float(loss))
    if prev is not None:
        for k in cur:
            if k in prev and prev[k].shape == cur[k].shape and not torch.equal(prev[k], cur[k]):
                first_bad.setdefault(k, 0)
                first_bad[k] += 1
    prev = cur
print("losses", ["%.7f" % l for l in losses])
print("modules whose output differed between consecutive passes (first 12 in forward order):", list(first_bad.items())[:12])
