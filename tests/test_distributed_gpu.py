"""The multi-GPU code paths on ONE MI355X: a live 1-rank RCCL group makes every collective of the PS data plane
really run (reduce_scatter / all_to_all / all_gather / all_reduce on the comm stream, launched from the autograd
hooks, bf16 wire, in-place slices), and the result must be bitwise what the collective-free path leaves.  Plus the
numerics of the device-side embedding router against its framework-op reference."""
import copy
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from distributed_tensorflow_amd.ops import hip

    hip.lib()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def rccl(dev):
    import torch.distributed as dist

    from distributed_tensorflow_amd.parallel.comm import Comm

    comm = Comm.from_env(dev, backend="nccl", force=True)
    assert comm.live and dist.get_backend() == "nccl"
    yield comm
    comm.barrier()


@pytest.mark.parametrize("push", ["reduce_scatter", "all_to_all"])
def test_bert_clipped_adam_through_rccl_single_rank(dev, rccl, push):
    """BERT's optimizer clips by global norm: packs + pushes run from the hooks, sum-of-squares (with the all-to-all
    slab sum fused in), the scalar all-reduce, K2 apply and the all-gather after backward.  Bitwise == no-comm path."""
    from distributed_tensorflow_amd.models import bert
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    torch.manual_seed(0)
    m1 = bert.prepare(bert.build(layers=2, hidden=256, heads=4, intermediate=1024, vocab=3000), dev)
    m2 = copy.deepcopy(m1)
    batch = bert.synthetic_batch(4, device=dev, seed=1, seq_len=128, vocab=3000)
    opt = bert.default_optimizer()
    ps1 = ShardedPS(list(m1.named_parameters()), rccl, opt, bucket_mb=1, push=push)
    ps2 = ShardedPS(list(m2.named_parameters()), Comm.single(dev), opt, bucket_mb=1)
    assert ps1.gnorm_sq is not None and len(ps1.layout.buckets) > 3
    assert sorted({b.group for b in ps1.layout.buckets}) == [0, 1]        # decayed / not decayed
    for _ in range(3):
        bert.loss_fn(m1, batch).backward()
        assert any(ps1._launched), "hooks must have pushed buckets during backward"
        for p, q in zip(m1.parameters(), m2.parameters()):
            q.grad = p.grad.clone()
        ps1.step()
        ps2.step()
        torch.cuda.synchronize()
        assert torch.equal(ps1.gnorm_sq, ps2.gnorm_sq)
        assert torch.equal(ps1.master, ps2.master)
        assert torch.equal(ps1.slot_v, ps2.slot_v)
        assert torch.equal(ps1.param_flat, ps2.param_flat)
    ps1.remove_hooks()
    ps2.remove_hooks()


def test_forward_order_pull_prefetch_is_bitwise_the_global_join(dev, rccl):
    """``ShardedPS.attach``: the next forward waits per bucket (forward pre-hooks on the owner modules, the
    functionally-read embedding tables at the root) instead of joining the whole comm stream.  BERT's clipped path
    applies + pulls in forward layer order.  Must leave bitwise what the joined path leaves, must really have
    switched to per-bucket waits, and the root set must hold the bucket of ``word_embeddings`` (read by the root
    module's own forward, never through ``nn.Embedding.forward``)."""
    from distributed_tensorflow_amd.models import bert
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    torch.manual_seed(0)
    m1 = bert.prepare(bert.build(layers=2, hidden=256, heads=4, intermediate=1024, vocab=3000), dev)
    m2 = copy.deepcopy(m1)
    batch = bert.synthetic_batch(4, device=dev, seed=1, seq_len=128, vocab=3000)
    opt = bert.default_optimizer()
    ps1 = ShardedPS(list(m1.named_parameters()), rccl, opt, bucket_mb=1).attach(m1)
    ps2 = ShardedPS(list(m2.named_parameters()), rccl, opt, bucket_mb=1)
    assert ps1._attached is not None and len(ps1.layout.buckets) > 3
    for it in range(5):
        l1 = bert.loss_fn(m1, batch)
        l1.backward()
        ps1.step()
        l2 = bert.loss_fn(m2, batch)
        l2.backward()
        ps2.step()
        torch.cuda.synchronize()
        assert torch.equal(l1, l2), it
        assert torch.equal(ps1.param_flat, ps2.param_flat) and torch.equal(ps1.master, ps2.master), it
    assert ps1._prefetch and ps1.prefetch_waits >= 3 * len(ps1.layout.buckets)
    emb_bucket = ps1._slot_bucket[id(m1.word_embeddings.weight)]
    assert emb_bucket in ps1._attached["root_buckets"]
    ps1.remove_hooks()
    ps2.remove_hooks()


def test_pull_prefetch_with_lr_schedule_and_no_syncs_is_bitwise_the_global_join(dev, rccl):
    """Six ResNet-50 steps back to back (no device synchronisation in between, a new learning rate every step), per-
    bucket waits against the global join: bitwise the same master weights.  Regression test for two races the global
    join used to hide: (a) ``set_lr`` filled the rate on the compute stream while the previous step's tail-bucket apply
    was still queued on the comm stream; (b) ``p.grad = None`` handed the BatchNorm gradients' blocks to the next
    forward's first allocations before the comm stream had packed them (bench: loss 4.3307 instead of 4.6683)."""
    from distributed_tensorflow_amd.models import resnet
    from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS

    torch.manual_seed(0)
    m1 = resnet.prepare(resnet.build(), dev).train()
    m2 = copy.deepcopy(m1)
    batch = resnet.synthetic_batch(64, device=dev, seed=1)      # (batch 32: MIOpen's heuristic pick for the untuned
    # stride-2 3x3 layer accumulates with atomics -- the forward itself is not reproducible there)
    opt = OptimizerSpec(kind="momentum", lr=0.02, momentum=0.9, weight_decay=1e-4)

    def run(m, ps):
        for it in range(7):
            ps.set_lr(0.02 * (it + 1) / 7.0)
            resnet.loss_fn(m, batch).backward()
            ps.step()
        ps.flush()
        torch.cuda.synchronize()

    m3 = copy.deepcopy(m1)
    ps1 = ShardedPS(list(m1.named_parameters()), rccl, opt, bucket_mb=8).attach(m1)
    run(m1, ps1)
    assert ps1._prefetch and ps1.prefetch_waits > 0 and len(ps1.layout.buckets) >= 3
    ps2 = ShardedPS(list(m2.named_parameters()), rccl, opt, bucket_mb=8)
    run(m2, ps2)
    ps3 = ShardedPS(list(m3.named_parameters()), rccl, opt, bucket_mb=8)
    run(m3, ps3)
    assert torch.isfinite(ps2.master).all()
    assert torch.equal(ps2.master, ps3.master), "the joined path itself is not reproducible"
    bad = (ps1.master != ps2.master)
    assert not bool(bad.any()), "%d master weights differ, first at %d" % (int(bad.sum()), int(bad.nonzero()[0]))
    assert torch.equal(ps1.param_flat, ps2.param_flat)
    for ps in (ps1, ps2, ps3):
        ps.remove_hooks()


def test_stale2_through_rccl_single_rank(dev, rccl):
    from distributed_tensorflow_amd.models import resnet
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    torch.manual_seed(0)
    m1 = resnet.prepare(resnet.build(), dev)
    m2 = copy.deepcopy(m1)
    batch = resnet.synthetic_batch(4, device=dev, seed=1, image_size=64)
    ps1 = ShardedPS(list(m1.named_parameters()), rccl, resnet.default_optimizer(), update_mode="stale", staleness=2)
    ps2 = ShardedPS(list(m2.named_parameters()), Comm.single(dev), resnet.default_optimizer(), update_mode="stale",
                    staleness=2)
    init = ps1.param_flat.clone()
    for it in range(4):
        resnet.loss_fn(m1, batch).backward()
        for p, q in zip(m1.parameters(), m2.parameters()):
            q.grad = p.grad.clone()
        ps1.step()
        ps2.step()
        torch.cuda.synchronize()
        if it < 2:
            assert torch.equal(ps1.param_flat, init), "updates must stay invisible for two steps"
        assert torch.equal(ps1.param_flat, ps2.param_flat)
    assert not torch.equal(ps1.param_flat, init)
    ps1.flush()
    ps2.flush()
    assert torch.equal(ps1.param_flat, ps2.param_flat) and torch.equal(ps1.master, ps2.master)


@pytest.mark.parametrize("partition", ["mod", "div"])
def test_wide_deep_routing_through_rccl_single_rank(dev, rccl, partition):
    """ShardedEmbedding honours a live 1-rank group: route kernel + three all-to-alls through RCCL, rows back through
    the slots, gradients to the owner -- tables and accumulators bitwise equal to the direct gather / apply."""
    from distributed_tensorflow_amd.models import wide_deep as wd
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    kw = dict(num_features=6, rows_per_table=5000, embedding_dim=16)
    torch.manual_seed(0)
    m1 = wd.prepare(wd.build(comm=rccl, device=dev, partition=partition, **kw), dev)
    torch.manual_seed(0)
    m2 = wd.prepare(wd.build(comm=Comm.single(dev), device=dev, **kw), dev)
    m2.embeddings.table.copy_(m1.embeddings.table)
    ps1 = ShardedPS(list(m1.named_parameters()), rccl, wd.default_optimizer())
    ps2 = ShardedPS(list(m2.named_parameters()), Comm.single(dev), wd.default_optimizer())
    batch = wd.synthetic_batch(4096, device=dev, seed=3, model=m1)
    for _ in range(5):          # default route "auto": two calibration lookups on the exact route, then three routed
        l1 = wd.loss_fn(m1, batch)
        l1.backward()
        ps1.step()
        l2 = wd.loss_fn(m2, batch)
        l2.backward()
        ps2.step()
        torch.cuda.synchronize()
        assert torch.equal(l1, l2)
        assert torch.equal(m1.embeddings.table, m2.embeddings.table)
        assert torch.equal(m1.embeddings.accumulator, m2.embeddings.accumulator)
    assert m1.embeddings.check_overflow() == 0 and m1.embeddings._lookups == 3 and m1.embeddings._calibrating == 0


@pytest.mark.parametrize("n,W,cap,div", [(1, 1, 1, False), (1000, 2, 600, False), (5000, 8, 700, False),
                                         (5000, 8, 500, False), (4097, 3, 2000, True), (70000, 8, 9000, True),
                                         (300000, 16, 20000, False), (1024, 4, 256, False)])
def test_route_kernel_matches_reference(dev, n, W, cap, div):
    """Stable bucket-by-owner split: slot of request i = owner * cap + (number of earlier requests of that owner);
    requests beyond the capacity are dropped and counted; padding ids (negative / out of range) get slot -1."""
    from distributed_tensorflow_amd.ops import embedding as E

    g = torch.Generator().manual_seed(n + W)
    total = 1000003
    ids = (torch.rand(n, generator=g) ** 3 * total).long()
    if n > 10:
        ids[::97] = -1
        ids[5] = total + 3
    rows_per = (total + W - 1) // W
    ref = E.route_reference(ids, W, cap, div, rows_per, total)
    ovf = torch.zeros(1, dtype=torch.int64, device=dev)
    got = E.embedding_route(ids.to(dev), W, cap, div, rows_per, total, ovf)
    torch.cuda.synchronize()
    for name, a, b in zip(("send_ids", "slot", "inv", "totals"), got, ref[:4]):
        assert torch.equal(a.cpu(), b), name
    assert int(ovf) == int(ref[4])
    if cap * W < n:
        assert int(ovf) > 0


@pytest.mark.parametrize("nsrc", [1, 2, 8])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_sumsq_fused_slab_sum(dev, nsrc, dtype):
    from distributed_tensorflow_amd.ops import reduce

    n, stride = 100003, 100008
    torch.manual_seed(nsrc)
    x = torch.randn(nsrc * stride, device=dev).to(dtype)
    out = torch.zeros(1, device=dev)
    reduce.sumsq(x, out, n=n, nsrc=nsrc, src_stride=stride)
    acc = sum(x[s * stride:s * stride + n].float() for s in range(nsrc))
    ref = (acc.double() ** 2).sum()
    assert abs(float(out) - float(ref)) <= 1e-5 * float(ref)
    reduce.sumsq(x, out, accumulate=True, n=n, nsrc=nsrc, src_stride=stride)
    assert abs(float(out) - 2 * float(ref)) <= 1e-5 * 2 * float(ref)


@pytest.mark.parametrize("name", ["mnist_mlp", "bert"])
def test_whole_step_hipgraph_with_live_rccl_is_bitwise_eager(dev, rccl, name):
    """The whole step -- forward, backward, the hook-launched PS pipeline with its RCCL collectives on the comm
    stream -- captured in ONE hipGraph and replayed must leave bitwise what the eager step leaves."""
    import time

    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    mod = get_model_module(name)
    torch.manual_seed(0)
    kw = dict(layers=2, hidden=256, heads=4, intermediate=1024, vocab=3000) if name == "bert" else {}
    m1 = mod.prepare(mod.build(**kw), dev) if hasattr(mod, "prepare") else mod.build(**kw).to(dev)
    m2 = copy.deepcopy(m1)
    bkw = dict(seq_len=128, vocab=3000) if name == "bert" else {}
    batch = mod.synthetic_batch(8, device=dev, seed=1, **bkw)
    opt = mod.default_optimizer()
    ps1 = ShardedPS(list(m1.named_parameters()), rccl, opt, bucket_mb=0.05)
    ps2 = ShardedPS(list(m2.named_parameters()), rccl, opt, bucket_mb=0.05)
    assert len(ps1.layout.buckets) >= 2

    def step(m, ps):
        loss = mod.loss_fn(m, batch)
        loss.backward()
        ps.step()
        return loss

    # both models take 3 eager steps (graph capture needs warm kernels / allocator / RCCL channels)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step(m1, ps1)
            step(m2, ps2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert torch.equal(ps1.param_flat, ps2.param_flat)
    time.sleep(0.5)                       # let the RCCL watchdog retire the warm-up collectives
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        static_loss = step(m2, ps2)       # capture runs nothing: m2 is one step behind until the first replay
    losses = []
    for _ in range(4):
        l1 = step(m1, ps1)
        g.replay()
        torch.cuda.synchronize()
        losses.append(float(l1))
        assert torch.equal(l1, static_loss)
        assert torch.equal(ps1.master, ps2.master)
        assert torch.equal(ps1.param_flat, ps2.param_flat)
    assert losses[-1] < losses[0]
    ps1.remove_hooks()
    ps2.remove_hooks()


def test_one_sided_async_ps_two_workers_share_one_gpu(dev, tmp_path):
    """The asynchronous PS on device memory: two worker PROCESSES on this one GPU (RCCL would refuse that; the mode
    needs no collective), each hosting one shard in hipMalloc memory that the other maps through HIP IPC and
    updates with the K1 kernel.  Workers take turns here, so the result must equal sequential SGD exactly."""
    from test_distributed import _sequential_reference, run_ranks

    res = run_ranks(2, "asyncserial_momentum_gpu", tmp_path, timeout=240)
    ref = _sequential_reference(2, "momentum")
    assert res["applies"] == 6
    for n, p in ref.items():
        torch.testing.assert_close(res["params"][n], p, rtol=1e-5, atol=1e-6, msg=lambda m: "%s: %s" % (n, m))


def test_one_sided_async_ps_free_running_on_gpu(dev, tmp_path):
    from test_distributed import run_ranks

    res = run_ranks(2, "asyncfree_momentum_gpu", tmp_path, timeout=240)
    fast, slow = res["steps"]
    assert fast > 2 * slow, res["steps"]
    assert res["losses"][-1] < 0.5 * res["losses"][0]


def _free_port():
    from conftest import free_port          # outside the kernel's ephemeral range, see there

    return free_port()


def test_bench_ends_at_the_same_loss_through_every_data_plane_path():
    """bench.py, ResNet-50 batch 64, twelve steps with the lr warm-up schedule, as separate processes: without a
    communicator (twice: run-to-run reproducibility), through a live 1-rank RCCL group with the pull prefetch, with the
    all-to-all push, with the native communicator, and through the one-sided async runtime -- one final loss, digit for
    digit.  This comparison is what exposed the K6T hazard and the two prefetch races of round 3
    (profiles/r3_path_consistency.txt)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env, *extra):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "8", "--warmup", "4", "--batch", "64",
               "--lr-warmup-steps", "10"] + list(extra)
        e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        e.update(env)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=e)
        assert r.returncode == 0, r.stderr[-3000:]
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        return out["config"]["final_loss"], out["config"]["first_loss"]

    ref = run({})
    assert ref[0] == ref[0] and 0.0 < ref[0] < 20.0          # finite (lr 0.1 after 10 steps: the loss may still hump)
    for name, env, extra in [("plain again", {}, []), ("live", {"DTF_FORCE_COMM": "1"}, []),
                             ("live all-to-all", {"DTF_FORCE_COMM": "1"}, ["--push", "all_to_all"]),
                             ("native", {"DTF_FORCE_COMM": "1", "DTF_COMM_BACKEND": "rccl-native"}, []),
                             ("async", {"DTF_FORCE_COMM": "1"}, ["--update-mode", "async"])]:
        got = run(env, *extra)
        assert got == ref, "%s: %r vs %r" % (name, got, ref)


@pytest.mark.parametrize("model,batch,steps", [("resnet50", 64, 8), ("bert", 16, 6), ("wide_deep", 65536, 6)])
def test_pipelined_step_equals_serialised_execution_bit_for_bit(dev, model, batch, steps):
    """The pipelined step (weight-gradient side stream, comm stream through a live 1-rank RCCL group, per-bucket pull
    prefetch, gradient hooks, no host synchronisation) against a fully serialised execution of the same training
    (AMD_SERIALIZE_KERNEL=3, synchronise after backward and after the PS step, no side stream, no prefetch), with a
    DIFFERENT batch and learning rate every step -- bench.py's single fixed batch would hide a consumer that reads a
    buffer one step early.  SHA-256 over every parameter, BatchNorm buffer, fp32 master and optimizer slot must be equal
    (scripts/serial_vs_pipelined.py, own processes)."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "serial_vs_pipelined.py")
    base = {k: v for k, v in os.environ.items() if not k.startswith("DTF_")}
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    hashes = []
    for extra in ({}, {"DTF_SERIAL": "1", "DTF_WGRAD_STREAM": "0", "DTF_PULL_PREFETCH": "0", "AMD_SERIALIZE_KERNEL": "3"}):
        env = dict(base)
        env.update(extra)
        env["MASTER_PORT"] = str(_free_port())
        r = subprocess.run([sys.executable, script, model, str(batch), str(steps)], capture_output=True, text=True, timeout=600,
                           env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        m = re.search(r"state hash over (\d+) tensors: ([0-9a-f]{64})", r.stdout)
        assert m, r.stdout[-1500:]
        losses = re.search(r"losses (.*)", r.stdout).group(1)
        hashes.append((m.group(2), losses))
    assert hashes[0][1] == hashes[1][1], hashes
    assert hashes[0][0] == hashes[1][0], "pipelined and serialised runs end in different states"
