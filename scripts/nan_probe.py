"""Which ResNet-50 gradients are non-finite after one forward + backward outside the PS runtime?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import resnet

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = resnet.prepare(resnet.build(), dev).train()
batch = resnet.synthetic_batch(B, device=dev, seed=1)
for it in range(3):
    for p in m.parameters():
        p.grad = None
    loss = resnet.loss_fn(m, batch)
    loss.backward()
    torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in m.named_parameters() if not torch.isfinite(p.grad).all()]
    print("iteration %d loss %.5f non-finite gradients: %s" % (it, float(loss), bad), flush=True)
