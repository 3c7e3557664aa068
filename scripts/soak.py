"""Soak: N steps of ResNet-50 through the live 1-rank PS pipeline (prefetch, side stream, lr schedule); device memory
and host RSS must stay flat, the loss finite."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import resnet
from distributed_tensorflow_amd.parallel.comm import Comm
from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
comm = Comm.from_env(force=True)
dev = comm.device
torch.manual_seed(0)
m = resnet.prepare(resnet.build(), dev).train()
batch = resnet.synthetic_batch(256, device=dev, seed=1)
opt = resnet.default_optimizer()
ps = ShardedPS(list(m.named_parameters()), comm, opt).attach(m)
for it in range(N):
    ps.set_lr(0.02 * min(1.0, (it + 1) / 200.0))
    l = resnet.loss_fn(m, batch); l.backward(); ps.step()
    if it % 250 == 0 or it == N - 1:
        torch.cuda.synchronize()
        print("step %5d loss %.4f  allocated %.1f MB reserved %.1f MB  host RSS %.0f MB" % (
            it, float(l), torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20,
            resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), flush=True)
ps.flush(); ps.remove_hooks(); comm.close()
