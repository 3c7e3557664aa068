"""Worker body for the multi-process gloo tests (spawned by torchrun-like launcher in test_distributed.py)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.models import mnist_mlp  # noqa: E402
from distributed_tensorflow_amd.parallel import checkpoint as ckpt  # noqa: E402
from distributed_tensorflow_amd.parallel.comm import Comm  # noqa: E402
from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS  # noqa: E402


def main():
    case, out_dir = sys.argv[1], sys.argv[2]
    comm = Comm.from_env(torch.device("cpu"), backend="gloo")
    r, W = comm.rank, comm.world_size
    torch.manual_seed(7)   # same init everywhere; broadcast_init also enforces it
    model = mnist_mlp.build()
    kind = "adam" if "adam" in case else "momentum"
    opt = OptimizerSpec(kind=kind, lr=0.01 if kind == "adam" else 0.1, momentum=0.9, weight_decay=1e-4,
                        max_grad_norm=1.0 if "clip" in case else 0.0)
    push = "all_to_all" if "a2a" in case else "reduce_scatter"
    mode = "stale" if "stale" in case else "sync"
    ps = ShardedPS(list(model.named_parameters()), comm, opt, bucket_mb=0.02, wire_dtype=torch.float32,
                   update_mode=mode, push=push)
    # the global batch is fixed (64); rank r trains on its 64/W slice
    x, y = mnist_mlp.synthetic_batch(64, seed=11)
    per = 64 // W
    batch = (x[r * per:(r + 1) * per], y[r * per:(r + 1) * per])
    steps = 6
    start = 0
    if case.startswith("resume"):
        rd = ckpt.restore_ps(ps, os.path.join(out_dir, "ckpt"), tf_names=mnist_mlp.tf_variable_names(model))
        start = rd.step
    losses = []
    for it in range(start, steps):
        loss = mnist_mlp.loss_fn(model, batch)
        loss.backward()
        ps.step()
        losses.append(float(loss.detach()))
        if case.startswith("save") and it + 1 == 3:
            ckpt.save_ps(ps, os.path.join(out_dir, "ckpt"), tf_names=mnist_mlp.tf_variable_names(model),
                         replicated={"global_step": torch.tensor(ps.global_step)} if r == 0 else None)
    ps.flush()
    full = ps.full_master()
    if r == 0:
        params = {n: ps.param_from_flat(full, i).clone() for i, (n, _) in enumerate(ps.named_params)}
        torch.save({"params": params, "losses": losses, "buckets": len(ps.layout.buckets)},
                   os.path.join(out_dir, "result_%s_w%d.pt" % (case, W)))
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
