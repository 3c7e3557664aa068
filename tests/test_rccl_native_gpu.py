"""The native RCCL communicator (``native/rccl_comm.cpp`` through ``parallel/rccl_native.py``) on one MI355X: a
1-rank communicator is a real RCCL communicator, so every entry point runs through ``ncclCommInitRank`` and RCCL's
kernels.  Numerics: each collective against its definition at W = 1; the PS step on it bitwise against the
collective-free path; the whole step captured in a hipGraph WITHOUT the watchdog drain the c10d path needs."""
import copy
import os

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
def native(dev):
    from distributed_tensorflow_amd.parallel.rccl_native import NativeComm, lib

    assert lib().dtf_rccl_abi_version() >= 1
    env = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE")}
    os.environ.update(RANK="0", WORLD_SIZE="1")
    comm = NativeComm.from_env(dev)
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    assert comm.live and comm.world_size == 1 and comm.backend == "rccl-native"
    yield comm
    comm.barrier()
    comm.close()
    assert not comm.live


def test_library_loaded_and_version(native):
    import ctypes

    from distributed_tensorflow_amd.parallel.rccl_native import lib

    v = ctypes.c_int()
    assert lib().dtf_rccl_version(ctypes.byref(v)) == 0 and v.value > 20000
    st = native.stats()
    assert st["nranks"] == 1 and st["rank"] == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_every_collective_at_one_rank(dev, native, dtype):
    torch.manual_seed(0)
    n = 1 << 16
    x = torch.randn(n, device=dev).to(dtype)
    before = native.stats()["collectives"]
    out = torch.zeros_like(x)
    native.reduce_scatter(out, x)
    assert torch.equal(out, x)
    out.zero_()
    native.all_gather(out, x)
    assert torch.equal(out, x)
    out.zero_()
    native.all_to_all(out, x)
    assert torch.equal(out, x)
    y = x.clone()
    native.all_reduce(y)
    assert torch.equal(y, x)
    native.all_reduce(y, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(y, x)
    native.broadcast(y, 0)
    native.reduce(y, 0)
    assert torch.equal(y, x)
    # variable splits, 2-D rows, and the empty exchange
    rows = torch.randn(37, 24, device=dev).to(dtype)
    back = torch.zeros_like(rows)
    native.all_to_all(back, rows, [37], [37])
    assert torch.equal(back, rows)
    e_in, e_out = torch.empty(0, 24, device=dev, dtype=dtype), torch.empty(0, 24, device=dev, dtype=dtype)
    native.all_to_all(e_out, e_in, [0], [0])
    ids = torch.arange(1000, device=dev)
    got = torch.zeros_like(ids)
    native.all_to_all(got, ids)
    assert torch.equal(got, ids)
    native.barrier()
    assert native.stats()["collectives"] >= before + 10 and native.healthy()


def test_in_place_slices_on_a_side_stream(dev, native):
    """What ShardedPS does: collectives on slices of flat buffers, issued on the comm stream."""
    flat = torch.randn(1 << 20, device=dev).to(torch.bfloat16)
    shard = torch.zeros(1 << 18, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        native.reduce_scatter(shard, flat[1 << 18:1 << 19])
        native.all_gather(flat[0:1 << 18], shard)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert torch.equal(flat[0:1 << 18], flat[1 << 18:1 << 19])


@pytest.mark.parametrize("push", ["reduce_scatter", "all_to_all"])
def test_ps_step_on_native_comm_is_bitwise_the_collective_free_path(dev, native, push):
    from distributed_tensorflow_amd.models import bert
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    torch.manual_seed(0)
    m1 = bert.prepare(bert.build(layers=2, hidden=256, heads=4, intermediate=1024, vocab=3000), dev)
    m2 = copy.deepcopy(m1)
    batch = bert.synthetic_batch(4, device=dev, seed=1, seq_len=128, vocab=3000)
    opt = bert.default_optimizer()
    ps1 = ShardedPS(list(m1.named_parameters()), native, opt, bucket_mb=1, push=push).attach(m1)
    ps2 = ShardedPS(list(m2.named_parameters()), Comm.single(dev), opt, bucket_mb=1)
    calls0 = native.stats()["collectives"]
    for _ in range(4):
        l1 = bert.loss_fn(m1, batch)
        l1.backward()
        ps1.step()
        l2 = bert.loss_fn(m2, batch)
        l2.backward()
        ps2.step()
        torch.cuda.synchronize()
        assert torch.equal(l1, l2)
        assert torch.equal(ps1.master, ps2.master) and torch.equal(ps1.param_flat, ps2.param_flat)
    nb = len(ps1.layout.buckets)
    assert native.stats()["collectives"] >= calls0 + 4 * (2 * nb + 1)      # push + pull per bucket + the norm
    ps1.remove_hooks()
    ps2.remove_hooks()


def test_whole_step_hipgraph_on_native_comm_needs_no_watchdog_drain(dev, native):
    """Capture right after the warm-up steps: no sleep, global capture mode (the c10d path needs thread-local mode
    and a 0.5 s drain of its watchdog's event queries)."""
    from distributed_tensorflow_amd.models import bert
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    torch.manual_seed(0)
    m1 = bert.prepare(bert.build(layers=2, hidden=256, heads=4, intermediate=1024, vocab=3000), dev)
    m2 = copy.deepcopy(m1)
    batch = bert.synthetic_batch(8, device=dev, seed=1, seq_len=128, vocab=3000)
    ps1 = ShardedPS(list(m1.named_parameters()), native, bert.default_optimizer(), bucket_mb=0.5)
    ps2 = ShardedPS(list(m2.named_parameters()), native, bert.default_optimizer(), bucket_mb=0.5)

    def step(m, ps):
        loss = bert.loss_fn(m, batch)
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
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = step(m2, ps2)
    for _ in range(3):
        l1 = step(m1, ps1)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(l1, static_loss)
        assert torch.equal(ps1.master, ps2.master) and torch.equal(ps1.param_flat, ps2.param_flat)
    ps1.remove_hooks()
    ps2.remove_hooks()
