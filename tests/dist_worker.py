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


def widedeep_case(comm, out_dir, case="widedeep", device=None):
    """Row-sharded embeddings over W ranks == the same model on one rank with the whole batch.
    Variants: ``_div`` (contiguous-range partition), ``_exact`` (variable-split exchange), ``_tight`` (capacity
    factor 1.0: overflow must be counted and reported), ``_save`` / ``_resume[_div]`` (checkpoint at step 3, resume on
    another world size / partition strategy)."""
    from distributed_tensorflow_amd.models import wide_deep as wd
    from distributed_tensorflow_amd.train import trainer

    r, W = comm.rank, comm.world_size
    device = device or torch.device("cpu")
    gpu = device.type == "cuda"
    torch.manual_seed(5)
    kw = dict(num_features=4, rows_per_table=97, embedding_dim=16 if gpu else 8, emb_optimizer="adagrad", emb_lr=0.1)
    model = wd.build(comm=comm, device=device, partition="div" if "div" in case else "mod", **kw)
    if gpu:
        model = wd.prepare(model, device)
    emb = model.embeddings
    if "exact" in case:
        emb.route, emb._calibrating = "exact", 0
    if "capacity" in case or "tight" in case:
        emb.route, emb._calibrating = "capacity", 0     # the lossy opt-in route from the first lookup
    if "tight" in case:
        emb.capacity_factor, emb.overflow_check_every = 0.5, 0
    # W-independent table init: the same global rows everywhere
    g = torch.Generator().manual_seed(99)
    rows = torch.arange(emb.total_rows)
    vals = torch.randn(emb.total_rows, emb.dim, generator=g) * 0.05
    emb.load_rows(rows.to(device), vals.to(device))
    ps = ShardedPS(list(model.named_parameters()), comm, OptimizerSpec(kind="momentum", lr=0.05, momentum=0.9),
                   bucket_mb=0.01, wire_dtype=None if gpu else torch.float32)
    tf_names = wd.tf_variable_names(model)
    B = 4096 if gpu else 64             # the fused bf16 tower wants GEMM-sized batches
    ids, dense, y = wd.synthetic_batch(B, seed=21, model=model)
    per = B // W
    sl = slice(r * per, (r + 1) * per)
    batch = tuple(t[sl].to(device) for t in (ids, dense, y))
    losses = []
    start, steps = 0, 5
    ck = os.path.join(out_dir, "ckpt_wd")
    if "resume" in case:
        start = trainer.restore_checkpoint(model, ps, ck, tf_names)
    for it in range(start, steps):
        loss = wd.loss_fn(model, batch)
        loss.backward()
        ps.step()
        losses.append(float(loss.detach()))
        if "save" in case and it + 1 == 3:
            trainer.save_checkpoint(model, ps, ck, it + 1, tf_names, r == 0)
    dropped = emb.check_overflow(grow=False) if emb.route != "exact" else 0
    # reassemble the full table on every rank
    full = torch.zeros(emb.total_rows, emb.dim, device=device)
    acc = torch.zeros(emb.total_rows, emb.dim, device=device)
    mine = emb.global_rows_of(r).to(device)
    full[mine] = emb.table.float()
    acc[mine] = emb.accumulator
    import torch.distributed as dist
    if W > 1:
        dist.all_reduce(full)
        dist.all_reduce(acc)
    dense_full = ps.full_master().cpu()
    full, acc = full.cpu(), acc.cpu()
    if r == 0:
        torch.save({"table": full, "acc": acc, "dense": dense_full, "losses": losses, "dropped": dropped},
                   os.path.join(out_dir, "result_%s_w%d.pt" % (case, W)))
    comm.barrier()
    comm.close()


def embedding_grow_case(comm, out_dir, case, device=None):
    """the overflow counter lives on the SENDING rank; a poll that lets only the overflowing
    ranks grow their factor leaves the ranks with different equal-split all-to-all sizes.  Rank 0's ids all belong
    to owner 0 (it overflows), the other ranks' ids are uniform (they do not): every poll must still leave one
    agreed factor everywhere, the exchange must keep working, and once the capacity fits the lookups are exact.
    ``_auto``: the default route -- calibrated on the exact route, it must never drop anything here."""
    import torch.distributed as dist

    from distributed_tensorflow_amd.parallel.embedding import ShardedEmbedding

    r, W = comm.rank, comm.world_size
    device = device or torch.device("cpu")
    auto = "auto" in case
    emb = ShardedEmbedding([4096], 16, comm, device=device, optimizer="sgd", lr=0.0, out_dtype=torch.float32,
                           route="auto" if auto else "capacity", capacity_factor=0.6, overflow_check_every=200)
    g = torch.Generator().manual_seed(3)
    ref = torch.randn(4096, 16, generator=g).to(device)
    emb.load_rows(torch.arange(4096, device=device), ref)
    gi = torch.Generator().manual_seed(10 + r)
    ids = torch.randint(0, 4096, (512,), generator=gi)
    if r == 0:
        ids = ids // W * W              # every request of rank 0 goes to owner 0
    ids = ids.to(device)
    factors, exact_after, dropped_total = [], None, 0
    for it in range(12):
        out = emb.lookup_flat(ids)
        out.sum().backward()
        f = torch.tensor([emb.capacity_factor], dtype=torch.float64, device=device)
        allf = [torch.zeros_like(f) for _ in range(W)]
        dist.all_gather(allf, f)
        assert all(float(x) == float(allf[0]) for x in allf), "capacity factors diverged: %s" % allf
        factors.append(float(f))
        ok = torch.tensor([float(torch.equal(out.detach(), ref[ids]))], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok) == 1.0 and exact_after is None:
            exact_after = it
        elif float(ok) != 1.0:
            exact_after = None
    dropped_total = emb.dropped_total
    t = torch.tensor([float(dropped_total)], device=device)
    dist.all_reduce(t)
    if r == 0:
        torch.save({"factors": factors, "exact_after": exact_after, "dropped": int(t)},
                   os.path.join(out_dir, "result_%s_w%d.pt" % (case, W)))
    comm.barrier()
    comm.close()


def real_layout_case(comm, out_dir, case, device=None):
    """The shipped ResNet-50 / BERT-base variable layouts (tail bucket, several 32 MB buckets, decay groups)
    with the bf16 wire, hooks launching the pipeline during backward, over W gloo ranks: every replica must hold
    bitwise the same parameters after each step, and the owned master shards must reassemble to them."""
    import torch.distributed as dist

    from distributed_tensorflow_amd.models import get_model_module

    r, W = comm.rank, comm.world_size
    name = "bert" if "bert" in case else "resnet50"
    mod = get_model_module(name)
    torch.manual_seed(3 + r)            # DIFFERENT init per rank: broadcast_init must make them equal
    device = device or torch.device("cpu")
    gpu = device.type == "cuda"
    model = mod.build()
    if gpu:
        model = mod.prepare(model, device)      # bf16 NHWC / fused graph: the parameters become views of the wire
    opt = mod.default_optimizer()
    push = "all_to_all" if "a2a" in case else "reduce_scatter"
    ps = ShardedPS(list(model.named_parameters()), comm, opt, wire_dtype=torch.bfloat16, push=push,
                   num_shards=1 if "ps1" in case else None)
    if gpu:         # shapes the hand-written kernels serve (K11 attention needs S in {128, ..}, the conv kernels 64^2)
        kw = dict(seq_len=128) if name == "bert" else dict(image_size=64)
        batch = mod.synthetic_batch(4, device=device, seed=50 + r, **kw)
    else:
        kw = dict(seq_len=16) if name == "bert" else dict(image_size=32)
        batch = mod.synthetic_batch(2, seed=50 + r, **kw)
    sums, losses = [], []
    for _ in range(2):
        loss = mod.loss_fn(model, batch)
        loss.backward()
        ps.step()
        losses.append(float(loss.detach()))
        # replicas in sync: compare an exact fingerprint of the pulled parameters across ranks
        fp = ps.param_flat.view(torch.int16).to(torch.int64)
        sig = torch.stack([fp.sum(), (fp * torch.arange(1, fp.numel() + 1, device=device) % 1000003).sum()])
        all_sig = [torch.zeros_like(sig) for _ in range(W)]
        dist.all_gather(all_sig, sig)
        assert all(torch.equal(all_sig[0], x) for x in all_sig), "replicas diverged: %s" % all_sig
        sums.append(int(sig[0]))
    full = ps.full_master()
    wire = ps.param_flat.float()
    # what every worker computes with is the bf16 rounding of the fp32 master the PS shards hold
    torch.testing.assert_close(full.to(torch.bfloat16).float(), wire, rtol=0, atol=0)
    # the model's own (fp32) tensors were refreshed from the pulled wire values
    for i, (n, p) in enumerate(ps.named_params):
        assert torch.equal(p.detach().reshape(-1), ps.param_from_flat(wire, i).reshape(-1).to(p.dtype)), n
    if r == 0:
        torch.save({"losses": losses, "buckets": len(ps.layout.buckets), "sums": sums,
                    "groups": sorted({b.group for b in ps.layout.buckets}),
                    "bucket_sizes": [b.size for b in ps.layout.buckets]},
                   os.path.join(out_dir, "result_%s_w%d.pt" % (case, W)))
    comm.barrier()
    comm.close()


def async_case(comm, out_dir, case, device=None):
    """One-sided asynchronous PS (parallel/async_ps.py).  ``asyncserial``: the workers take turns (a barrier per
    turn), so the result must equal applying their gradients one after the other -- exact.  ``asyncfree``: no
    synchronisation at all, worker 1 sleeps between steps: worker 0 must NOT be slowed down by it (its step count
    races ahead), every apply is counted by the owners, and the loss still falls."""
    import time

    import torch.distributed as dist

    from distributed_tensorflow_amd.parallel.async_ps import AsyncPS

    r, W = comm.rank, comm.world_size
    device = device or torch.device("cpu")
    torch.manual_seed(7)
    model = mnist_mlp.build().to(device)
    kind = "adam" if "adam" in case else ("sgd" if "sgd" in case else "momentum")
    opt = OptimizerSpec(kind=kind, lr=0.002 if kind == "adam" else 0.05, momentum=0.9)
    ps = AsyncPS(list(model.named_parameters()), comm, opt, bucket_mb=0.02, wire_dtype=torch.float32)
    x, y = mnist_mlp.synthetic_batch(64, seed=11)
    per = 64 // W
    batch = (x[r * per:(r + 1) * per].to(device), y[r * per:(r + 1) * per].to(device))
    losses, steps_done = [], 0
    if "serial" in case:
        for it in range(3):
            for turn in range(W):
                if turn == r:
                    ps.pull()
                    loss = mnist_mlp.loss_fn(model, batch)
                    loss.backward()
                    ps.step()
                    ps.flush()
                    losses.append(float(loss.detach()))
                    steps_done += 1
                dist.barrier()
    else:
        t_end = time.time() + 1.5
        while time.time() < t_end:
            loss = mnist_mlp.loss_fn(model, batch)
            loss.backward()
            ps.step()
            losses.append(float(loss.detach()))
            steps_done += 1
            if r == 1:
                time.sleep(0.05)        # a slow worker
    ps.finish()
    counts = torch.tensor([float(steps_done)])
    all_counts = [torch.zeros(1) for _ in range(W)]
    dist.all_gather(all_counts, counts)
    applies = ps.applies_received()
    full = ps.full_master().cpu()
    if r == 0:
        params = {n: ps.param_from_flat(full, i).clone() for i, (n, _) in enumerate(ps.named_params)}
        torch.save({"params": params, "losses": losses, "steps": [int(c) for c in all_counts], "applies": applies,
                    "buckets": len(ps.layout.buckets)}, os.path.join(out_dir, "result_%s_w%d.pt" % (case, W)))
    comm.barrier()
    ps.close()
    comm.close()


def graph_case(comm, out_dir, case, device):
    """Whole-step hipGraph under a live W-rank RCCL group: model A steps eagerly, model B (same init, same batch)
    replays one captured step; after every step the two must be bitwise equal on every rank, and the replicas
    bitwise equal across ranks.  Both models issue their collectives in the same order on every rank."""
    import copy
    import time

    import torch.distributed as dist

    from distributed_tensorflow_amd.models import get_model_module

    assert device.type == "cuda"
    r, W = comm.rank, comm.world_size
    name = "bert" if "bert" in case else "mnist_mlp"
    mod = get_model_module(name)
    torch.manual_seed(0)
    kw = dict(layers=2, hidden=256, heads=4, intermediate=1024, vocab=3000) if name == "bert" else {}
    m1 = mod.prepare(mod.build(**kw), device) if hasattr(mod, "prepare") else mod.build(**kw).to(device)
    m2 = copy.deepcopy(m1)
    bkw = dict(seq_len=128, vocab=3000) if name == "bert" else {}
    batch = mod.synthetic_batch(8, device=device, seed=1 + r, **bkw)
    opt = mod.default_optimizer()
    ps1 = ShardedPS(list(m1.named_parameters()), comm, opt, bucket_mb=0.05)
    ps2 = ShardedPS(list(m2.named_parameters()), comm, opt, bucket_mb=0.05)

    def step(m, ps):
        loss = mod.loss_fn(m, batch)
        loss.backward()
        ps.step()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step(m1, ps1)
            step(m2, ps2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    assert torch.equal(ps1.param_flat, ps2.param_flat)
    time.sleep(0.5)                       # the RCCL watchdog retires the warm-up collectives
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        static_loss = step(m2, ps2)
    losses = []
    for _ in range(4):
        l1 = step(m1, ps1)
        g.replay()
        torch.cuda.synchronize()
        losses.append(float(l1))
        assert torch.equal(l1, static_loss), "replayed loss differs from eager"
        assert torch.equal(ps1.master, ps2.master) and torch.equal(ps1.param_flat, ps2.param_flat)
        fp = ps2.param_flat.view(torch.int16).to(torch.int64).sum().reshape(1)
        allfp = [torch.zeros_like(fp) for _ in range(W)]
        dist.all_gather(allfp, fp)
        assert all(torch.equal(allfp[0], x) for x in allfp), "replicas diverged under graph replay"
    if r == 0:
        torch.save({"losses": losses, "buckets": len(ps1.layout.buckets)},
                   os.path.join(out_dir, "result_%s_w%d.pt" % (case, W)))
    comm.barrier()
    comm.close()


def main():
    case, out_dir = sys.argv[1], sys.argv[2]
    # DTF_TEST_DEVICE=cuda: rank r on GPU r over RCCL (tests/test_multigpu.py); a 1-rank group is kept LIVE so the
    # same cases run through real collectives on a one-GPU box.  Default: CPU over gloo.
    if os.environ.get("DTF_TEST_DEVICE", "cpu") == "cuda" and not case.startswith("async"):
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        # DTF_TEST_COMM=rccl-native: direct RCCL calls (native/rccl_comm.cpp) instead of torch.distributed
        comm = Comm.from_env(device, backend=os.environ.get("DTF_TEST_COMM", "nccl"), force=True)
    else:
        device = torch.device("cpu")
        comm = Comm.from_env(device, backend="gloo")
    r, W = comm.rank, comm.world_size
    if case.startswith("async"):
        # _gpu: every worker on device 0 (two processes share ONE GPU); _multigpu: worker r on device r, the
        # one-sided applies then cross xGMI.  The group is gloo either way (control plane only).
        dev = None
        if case.endswith("_multigpu"):
            dev = torch.device("cuda", r % torch.cuda.device_count())
            torch.cuda.set_device(dev)
        elif case.endswith("_gpu"):
            dev = torch.device("cuda", 0)
        return async_case(comm, out_dir, case, dev)
    if case.startswith("widedeep"):
        return widedeep_case(comm, out_dir, case, device)
    if case.startswith("embgrow"):
        return embedding_grow_case(comm, out_dir, case, device)
    if case.startswith("real_"):
        return real_layout_case(comm, out_dir, case, device)
    if case.startswith("graph_"):
        return graph_case(comm, out_dir, case, device)
    torch.manual_seed(7)   # same init everywhere; broadcast_init also enforces it
    model = mnist_mlp.build().to(device)
    kind = "adam" if "adam" in case else "momentum"
    opt = OptimizerSpec(kind=kind, lr=0.01 if kind == "adam" else 0.1, momentum=0.9,
                        weight_decay=0.05 if "nodecay" in case else 1e-4,
                        max_grad_norm=1.0 if "clip" in case else 0.0,
                        no_decay=("_b",) if "nodecay" in case else ())
    push = "all_to_all" if "a2a" in case else "reduce_scatter"
    mode = "stale" if "stale" in case else "sync"
    # "tail": one big bucket + the small tail bucket of the first layer (hid_w + hid_b = 78 500 elements)
    sizes = dict(bucket_mb=1.0, tail_mb=0.3) if "tail" in case else dict(bucket_mb=0.02)
    ps = ShardedPS(list(model.named_parameters()), comm, opt, wire_dtype=torch.float32, update_mode=mode, push=push,
                   num_shards=1 if "ps1" in case else None, staleness=2 if "stale2" in case else 1, **sizes)
    # the global batch is fixed (64); rank r trains on its 64/W slice
    x, y = mnist_mlp.synthetic_batch(64, seed=11)
    per = 64 // W
    batch = (x[r * per:(r + 1) * per].to(device), y[r * per:(r + 1) * per].to(device))
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
    full = ps.full_master().cpu()
    if r == 0:
        params = {n: ps.param_from_flat(full, i).clone() for i, (n, _) in enumerate(ps.named_params)}
        torch.save({"params": params, "losses": losses, "buckets": len(ps.layout.buckets)},
                   os.path.join(out_dir, "result_%s_w%d.pt" % (case, W)))
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
