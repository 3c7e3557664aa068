"""Soak of the input pipeline: N steps of ResNet-50 (batch 128) fed from a memory-mapped uint8 file through the prefetcher;
host RSS, pinned / device memory must stay flat after the first steps.  usage: input_soak.py [steps]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
import numpy as np
import psutil
import torch
from distributed_tensorflow_amd.models import get_model_module
from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS
from distributed_tensorflow_amd.train import data as D

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(prefix="dtf-soak-")
rng = np.random.default_rng(0)
N = 2048
img = np.lib.format.open_memmap(os.path.join(d, "images.npy"), mode="w+", dtype=np.uint8, shape=(N, 224, 224, 3))
for i in range(0, N, 256):
    img[i:i + 256] = rng.integers(0, 256, (256, 224, 224, 3), dtype=np.uint8)
img.flush()
del img
np.save(os.path.join(d, "labels.npy"), rng.integers(0, 1000, N).astype(np.int64))
mod = get_model_module("resnet50")
torch.manual_seed(0)
m = mod.prepare(mod.build(), dev).train()
ps = ShardedPS(list(m.named_parameters()), None, mod.default_optimizer())
ps.set_lr(0.01)
ds = D.open_dataset(d)
pf = D.Prefetcher(D.BatchStream(ds, 128, seed=0), D.transform_for(mod, ds, dev), dev)
proc = psutil.Process()
marks = []
for it in range(steps):
    loss = mod.loss_fn(m, next(pf))
    loss.backward()
    ps.step()
    if it in (40, steps // 2, steps - 1):
        torch.cuda.synchronize()
        marks.append((it, proc.memory_info().rss >> 20, torch.cuda.memory_reserved() >> 20, torch.cuda.memory_allocated() >> 20, float(loss)))
        print("step %4d  host RSS %6d MB  device reserved %6d MB  allocated %6d MB  loss %.4f" % marks[-1], flush=True)
pf.close()
ok = marks[-1][1] - marks[0][1] < 256 and marks[-1][2] - marks[0][2] < 256
print("flat:", ok)
sys.exit(0 if ok else 1)
