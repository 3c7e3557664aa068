"""Loss trajectory of sync vs stale-1 vs stale-2 on one fixed ResNet-50 batch (one GPU, no communicator)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
import torch
from distributed_tensorflow_amd.models import resnet
from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for mode, k in (("sync", 0), ("stale", 1), ("stale", 2)):
    torch.manual_seed(0)
    m = resnet.prepare(resnet.build(), dev).train()
    batch = resnet.synthetic_batch(B, device=dev, seed=1)
    ps = ShardedPS(list(m.named_parameters()), None, OptimizerSpec(kind="momentum", lr=0.05, momentum=0.9),
                   update_mode=mode, staleness=max(k, 1))
    losses = []
    for it in range(30):
        ps.set_lr(0.05 * min(1.0, (it + 1) / 20.0))
        l = resnet.loss_fn(m, batch)
        l.backward()
        ps.step()
        losses.append(float(l))
    print(mode, k, " ".join("%.3f" % x for x in losses), flush=True)
