"""bench.py's training loop (default: ResNet-50, batch 256; argv: batch steps model), ShardedPS, with per-step loss and a finiteness check of gradients / parameters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import get_model_module
from distributed_tensorflow_amd.parallel.comm import Comm
from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

torch.manual_seed(1234)
comm = Comm.from_env()
dev = comm.device
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
resnet = get_model_module(sys.argv[3] if len(sys.argv) > 3 else "resnet50")      # any model module with the bench interface
m = resnet.build()
m = (resnet.prepare(m, dev) if hasattr(resnet, "prepare") else m.to(dev)).train()
ps = ShardedPS(list(m.named_parameters()), comm, resnet.default_optimizer(), bucket_mb=32.0)
batch = resnet.synthetic_batch(B, device=dev, seed=100)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    loss = resnet.loss_fn(m, batch)
    loss.backward()
    torch.cuda.synchronize()
    bad = [n for n, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    ps.step()
    torch.cuda.synchronize()
    badp = [n for n, p in m.named_parameters() if not torch.isfinite(p).all()]
    print("step %d loss %.5f non-finite gradients %d %s parameters %d" % (it, float(loss), len(bad), bad[:4], len(badp)), flush=True)
