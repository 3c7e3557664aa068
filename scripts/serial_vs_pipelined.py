"""Does the pipelined step (weight-gradient side stream, comm stream with per-bucket pull prefetch, hooks, no host
synchronisation anywhere) compute exactly what a fully serialised execution computes?

bench.py trains on ONE fixed batch, so a consumer that reads a buffer one step early sees almost the values it should
see, and a determinism check passes as long as the timing repeats.  Here every step gets a DIFFERENT batch and a different
learning rate; the run prints a hash over all parameters, BatchNorm buffers and optimizer state at the end.  Run it twice
-- as is, and with DTF_SERIAL=1 (synchronise after backward and after the PS step; combine with DTF_WGRAD_STREAM=0
DTF_PULL_PREFETCH=0 and AMD_SERIALIZE_KERNEL=3 for the strongest form) -- the two hashes must be equal bit for bit.
usage: serial_vs_pipelined.py [model] [batch] [steps]"""
import hashlib
import os
import sys

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

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
serial = os.environ.get("DTF_SERIAL", "0") == "1"
torch.manual_seed(1234)
comm = Comm.from_env(force=bool(int(os.environ.get("DTF_FORCE_COMM", "1"))), backend=os.environ.get("DTF_COMM_BACKEND") or None)
dev = comm.device
mod = get_model_module(name)
kw = dict(comm=comm, device=dev) if getattr(mod, "NEEDS_COMM", False) else {}
m = mod.build(**kw)
m = (mod.prepare(m, dev) if hasattr(mod, "prepare") else m.to(dev)).train()
opt = mod.default_optimizer()
mode = os.environ.get("DTF_UPDATE_MODE", "sync")               # sync | stale | async
if mode == "async":
    from distributed_tensorflow_amd.parallel.async_ps import AsyncPS

    ps = AsyncPS(list(m.named_parameters()), comm, opt, bucket_mb=32.0)
else:
    ps = ShardedPS(list(m.named_parameters()), comm, opt, bucket_mb=32.0, update_mode=mode,
                   push=os.environ.get("DTF_PUSH", "reduce_scatter"), staleness=int(os.environ.get("DTF_STALENESS", "1")))
if mode == "sync" and hasattr(ps, "attach"):
    ps.attach(m)
bkw = dict(model=m) if getattr(mod, "NEEDS_COMM", False) else {}
batches = [mod.synthetic_batch(B, device=dev, seed=500 + i, **bkw) for i in range(5)]       # a different batch every step
torch.cuda.synchronize()
losses = []
for it in range(steps):
    ps.set_lr(opt.lr * (0.2 + 0.8 * ((it * 7) % 5) / 4.0))                                  # and a different rate
    loss = mod.loss_fn(m, batches[it % len(batches)])
    loss.backward()
    if serial:
        torch.cuda.synchronize()
    ps.step()
    if serial:
        torch.cuda.synchronize()
    losses.append(loss.detach())
torch.cuda.synchronize()
h = hashlib.sha256()
n = 0
for k, v in sorted(m.state_dict().items()):
    h.update(k.encode())
    h.update(v.detach().cpu().reshape(-1).contiguous().view(torch.uint8).numpy().tobytes() if v.numel() else b"")
    n += 1
for t in (getattr(ps, "master", None), getattr(ps, "slot_m", None), getattr(ps, "slot_v", None)):
    if isinstance(t, torch.Tensor):
        h.update(t.detach().cpu().reshape(-1).contiguous().view(torch.uint8).numpy().tobytes())
print("%s batch %d steps %d mode=%s push=%s comm=%s serial=%d side_stream=%s prefetch=%s: losses %s" % (
    name, B, steps, mode, os.environ.get("DTF_PUSH", "reduce_scatter"), os.environ.get("DTF_COMM_BACKEND", "nccl"), serial, os.environ.get("DTF_WGRAD_STREAM", "1"), os.environ.get("DTF_PULL_PREFETCH", "1"),
    " ".join("%.4f" % float(l) for l in losses)))
print("state hash over %d tensors: %s" % (n, h.hexdigest()))
comm.close()
