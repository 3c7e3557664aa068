"""Forward + backward of ResNet-50 (batch 64) several times on the same input: every gradient must be bitwise equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import get_model_module

# usage: determinism_check.py [batch] [model]   (model: resnet50 (default), inception_v3, bert)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
resnet = get_model_module(sys.argv[2] if len(sys.argv) > 2 else "resnet50")
m = resnet.prepare(resnet.build(), dev).train()
batch = resnet.synthetic_batch(B, device=dev, seed=1)
ref = None
bad = 0
for it in range(6):
    for p in m.parameters():
        p.grad = None
    loss = resnet.loss_fn(m, batch)
    loss.backward()
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in m.named_parameters()}
    if ref is None:
        ref, ref_loss = g, float(loss)
        continue
    diff = [n for n in g if not torch.equal(g[n], ref[n])]
    if diff:
        bad += 1
        mags = ["%s %.2e" % (n, float((g[n].float() - ref[n].float()).norm() / ref[n].float().norm())) for n in diff[:8]]
        print("iteration %d vs %d: %d tensors differ (loss %.7f vs %.7f): %s" % (it, it - 1, len(diff), float(loss), ref_loss, "; ".join(mags)), flush=True)
    ref, ref_loss = g, float(loss)
from distributed_tensorflow_amd.ops import streams
print("loss %.6f  non-reproducible iterations: %d  (weight gradients copied into the parameter layout on the side stream: %d per pass)"
      % (float(loss), bad, streams.layout_copies // 6))
