"""One rank: AsyncPS and ShardedPS must leave bitwise the same parameters (same kernels, same order of operations)."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
import torch
from distributed_tensorflow_amd.models import resnet
from distributed_tensorflow_amd.parallel.async_ps import AsyncPS
from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m1 = resnet.prepare(resnet.build(), dev).train()
m2 = copy.deepcopy(m1)
batch = resnet.synthetic_batch(64, device=dev, seed=1)
opt = resnet.default_optimizer()
comm = None
if os.environ.get("LIVE", "0") == "1":
    from distributed_tensorflow_amd.parallel.comm import Comm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    comm = Comm.from_env(force=True)
a = AsyncPS(list(m1.named_parameters()), comm, opt, bucket_mb=32)
b = ShardedPS(list(m2.named_parameters()), comm, opt, bucket_mb=32)
if os.environ.get("NOSYNC", "0") == "1":
    # like the bench: all steps of one runtime back to back, no device synchronisation in between
    B = int(os.environ.get("BATCH", "256")); N = int(os.environ.get("STEPS", "30"))
    batch = resnet.synthetic_batch(B, device=dev, seed=1)
    for m, ps in ((m1, a), (m2, b)):
        for it in range(N):
            ps.set_lr(opt.lr * min(1.0, (it + 1) / 100.0))
            l = resnet.loss_fn(m, batch); l.backward(); ps.step()
        torch.cuda.synchronize()
        print(type(ps).__name__, "final loss %.6f" % float(l), flush=True)
    bad = [n for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()) if not torch.equal(p, q)]
    print("step", N, "params that differ:", len(bad), bad[:6], flush=True)
    if comm is not None:
        a.close(); comm.close()
    sys.exit(0)
sync_each = os.environ.get("SYNC_EACH", "0") == "1"
for it in range(6):
    for m, ps in ((m1, a), (m2, b)):
        ps.set_lr(opt.lr * (it + 1) / 6.0)
        l = resnet.loss_fn(m, batch); l.backward(); ps.step()
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    bad = [n for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()) if not torch.equal(p, q)]
    print("step", it, "params that differ:", len(bad), bad[:6], flush=True)

if comm is not None:
    a.close(); comm.close()
