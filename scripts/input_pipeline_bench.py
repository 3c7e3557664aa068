"""Does the input pipeline keep the GPU fed?  ResNet-50 batch 256 trained from a uint8 NHWC ``images.npy`` (memory-mapped,
random rows, pinned-host + copy-stream prefetch, normalisation on the device) against the same loop on one resident
synthetic batch.  usage: input_pipeline_bench.py [n_images] [steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import numpy as np
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import get_model_module
from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS
from distributed_tensorflow_amd.train import data as D
from distributed_tensorflow_amd.train.trainer import train_loop

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(prefix="dtf-data-")
rng = np.random.default_rng(0)
img = np.lib.format.open_memmap(os.path.join(d, "images.npy"), mode="w+", dtype=np.uint8, shape=(N, 224, 224, 3))
for i in range(0, N, 256):
    img[i:i + 256] = rng.integers(0, 256, (min(256, N - i), 224, 224, 3), dtype=np.uint8)
img.flush()
del img
np.save(os.path.join(d, "labels.npy"), rng.integers(0, 1000, N).astype(np.int64))
mod = get_model_module("resnet50")
res = {}
for how in ("synthetic", "dataset", "synthetic", "dataset"):
    torch.manual_seed(0)
    m = mod.prepare(mod.build(), dev).train()
    ps = ShardedPS(list(m.named_parameters()), None, mod.default_optimizer())
    ps.set_lr(0.01)
    batches = None
    if how == "dataset":
        ds = D.open_dataset(d)
        batches = D.Prefetcher(D.BatchStream(ds, 256, seed=0), D.transform_for(mod, ds, dev), dev)
        first = next(batches)
    else:
        first = mod.synthetic_batch(256, device=dev, seed=1)
    train_loop(m, mod, ps, first, 8, log_every=0, batches=batches)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_loop(m, mod, ps, first, steps, log_every=0, batches=batches)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if batches is not None:
        batches.close()
    res.setdefault(how, []).append(dt)
    print("%-9s %.2f ms/step  %.0f images/s" % (how, dt * 1e3, 256 / dt), flush=True)
    ps.remove_hooks()
    del m, ps
print("dataset / synthetic step time: %.3f" % (min(res["dataset"]) / min(res["synthetic"])))
