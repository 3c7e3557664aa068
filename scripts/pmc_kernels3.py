"""Third PMC driver (round 3): whole training steps instead of isolated calls, so every own kernel is counted at the
shapes the models give it -- ResNet-50 batch 256 (K13-K19, K7 family, stem), Wide&Deep batch 65536 (K3/K4 embedding,
K20 row split / merge, the restricted-bit sort) -- forward + backward, two iterations each (the first one warms caches).
The weight-gradient side stream is off: counters are collected kernel by kernel anyway."""
import os, sys
os.environ["DTF_WGRAD_STREAM"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import get_model_module

from distributed_tensorflow_amd.parallel.comm import Comm

dev = torch.device("cuda", 0)
comm = Comm.from_env()
which = sys.argv[1:] or ["resnet50", "wide_deep"]
for name in which:
    torch.manual_seed(0)
    mod = get_model_module(name)
    needs_comm = getattr(mod, "NEEDS_COMM", False)
    m = mod.prepare(mod.build(**(dict(comm=comm, device=dev) if needs_comm else {})), dev).train()
    B = {"resnet50": 256, "wide_deep": 65536, "inception_v3": 256, "bert": 64}[name]
    batch = mod.synthetic_batch(B, device=dev, seed=1, **(dict(model=m) if needs_comm else {}))
    for it in range(2):
        for p in m.parameters():
            p.grad = None
        loss = mod.loss_fn(m, batch)
        loss.backward()
        torch.cuda.synchronize()
    print(name, "loss %.4f" % float(loss), flush=True)
    del m, batch, loss
    torch.cuda.empty_cache()
print("done")
