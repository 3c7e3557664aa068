"""train_loop with graph = 0 / 1 / auto on ResNet-50 at a batch size: wall time per step in windows of 10 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
import torch
from distributed_tensorflow_amd.models import get_model_module
from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS
from distributed_tensorflow_amd.train.trainer import train_loop

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
graph = {"0": False, "1": True}.get(sys.argv[2] if len(sys.argv) > 2 else "auto", "auto")
dev = torch.device("cuda", 0)
mod = get_model_module("resnet50")
torch.manual_seed(0)
m = mod.prepare(mod.build(), dev).train()
ps = ShardedPS(list(m.named_parameters()), None, mod.default_optimizer())
ps.set_lr(0.01)
batch = mod.synthetic_batch(B, device=dev, seed=1)
marks = []


def on_step(step, loss):
    if step % 10 == 0:
        torch.cuda.synchronize()
        marks.append((step, time.perf_counter()))


train_loop(m, mod, ps, batch, 120, log_every=0, graph=graph, on_step=on_step)
print("batch %d graph=%s GPU_MAX_HW_QUEUES=%s:" % (B, graph, os.environ.get("GPU_MAX_HW_QUEUES")),
      " ".join("%d:%.1f" % (b[0], (b[1] - a[1]) / 10 * 1e3) for a, b in zip(marks, marks[1:])), "ms/step per window")
