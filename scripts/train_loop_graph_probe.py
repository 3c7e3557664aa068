"""train_loop(graph=True) on the GPU with mnist_mlp: argv[1] = fixed | cycle | prefetch (where the batches come from)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler
faulthandler.enable()
import numpy as np
import torch
from distributed_tensorflow_amd.models import get_model_module
from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS
from distributed_tensorflow_amd.train import data as D
from distributed_tensorflow_amd.train.trainer import train_loop

how = sys.argv[1] if len(sys.argv) > 1 else "fixed"
dev = torch.device("cuda", 0)
mod = get_model_module("mnist_mlp")
torch.manual_seed(0)
model = mod.build().to(dev)
ps = ShardedPS(list(model.named_parameters()), None, OptimizerSpec(kind="momentum", lr=0.05, momentum=0.9))
batches = None
if how == "fixed":
    first = mod.synthetic_batch(64, device=dev, seed=1)
elif how == "cycle":
    batches = D.Cycle([mod.synthetic_batch(64, device=dev, seed=s) for s in range(4)])
    first = next(batches)
else:
    rng = np.random.default_rng(0)
    y = rng.integers(0, 10, 2048).astype(np.uint8)
    x = (rng.random((2048, 784)) * 80).astype(np.uint8)
    for k in range(10):
        x[y == k, k * 78:(k + 1) * 78] += 150
    ds = D.ArrayDataset([x.reshape(-1, 28, 28), y], kind="mnist")
    batches = D.Prefetcher(D.BatchStream(ds, 64, seed=0), D.transform_for(mod, ds, dev), dev)
    first = next(batches)
losses = []
train_loop(model, mod, ps, first, 30, log_every=0, graph=True, batches=batches,
           on_step=lambda step, loss: losses.append(loss.detach().float().clone()))
print(how, "ok:", " ".join("%.3f" % float(v) for v in losses[::3]), flush=True)
