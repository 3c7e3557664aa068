"""The multi-GPU data plane on real devices: one process per GPU over RCCL / xGMI (SURVEY 4.1 'Distributed -- GPU').

Every test spawns W fresh worker processes (``tests/dist_worker.py`` with ``DTF_TEST_DEVICE=cuda``: rank r on GPU r,
backend ``nccl`` = RCCL) and needs ``torch.cuda.device_count() >= W``: on a one-GPU box the W >= 2 parametrisations
SKIP, on an 8-GPU node they all run.  The W = 1 parametrisations run everywhere: the same worker code on ``cuda:0``
through a live 1-rank RCCL group, so the device paths of every case are exercised even where no second GPU exists.

What is checked is what the reference's job model promises (``detail.ps`` PS tasks + ``detail.worker`` workers,
/root/reference/handlers/train_handler.py:96-112): sharded sync update == single-process training on the
concatenated batch (reduce-scatter and all-to-all push), replicas bitwise in sync with the bf16 wire on the real
ResNet-50 / BERT layouts (BERT = the clipped path), ``detail.ps < W`` (reduce / broadcast family), stale-2,
one-sided AsyncPS across devices (HIP IPC over xGMI), row-sharded embeddings routed by all-to-all == one rank, the
collective overflow poll, and the whole-step hipGraph under a W-rank group.
"""
import os
import subprocess
import sys

import pytest
import torch

from conftest import free_port

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _need(world):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    n = torch.cuda.device_count()           # does not initialise the devices
    if n < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, n))


def run_gpu_ranks(world, case, out_dir, timeout=420, backend="nccl", _retry=True):
    os.makedirs(out_dir, exist_ok=True)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DTF_TEST_DEVICE="cuda", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   OMP_NUM_THREADS="2", DTF_TEST_COMM=backend)
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "dist_worker.py"), case, str(out_dir)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    if _retry and any(p.returncode != 0 for p in procs) and any("EADDRINUSE" in o for o in outs):
        return run_gpu_ranks(world, case, out_dir, timeout, backend, _retry=False)      # somebody took the port: once more
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return torch.load(os.path.join(out_dir, "result_%s_w%d.pt" % (case, world)), weights_only=True)


WORLDS = [1, 2, 4, 8]


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("case", ["sync_momentum", "sync_momentum_a2a", "sync_adam_clip", "sync_momentum_clip_a2a",
                                  "sync_momentum_ps1", "sync_adam_clip_ps1"])
def test_sharded_ps_on_gpus_equals_single_process(tmp_path, case, world):
    """K5 pack -> RCCL push (reduce_scatter / all_to_all with the W-way sum fused into K1/K2 / rooted reduce for
    ``detail.ps = 1``) -> K1/K2 shard apply -> RCCL pull on W GPUs == torch.optim on the concatenated batch."""
    _need(world)
    from test_distributed import single_process_reference

    res = run_gpu_ranks(world, case, tmp_path)
    assert res["buckets"] >= 2
    ref, _ = single_process_reference("adam" if "adam" in case else "momentum", "clip" in case)
    for n, p in ref.items():
        torch.testing.assert_close(res["params"][n], p, rtol=5e-4, atol=5e-5, msg=lambda m: "%s: %s" % (n, m))
    assert res["losses"][-1] < res["losses"][0]


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("case", ["sync_momentum", "sync_momentum_a2a", "sync_adam_clip", "sync_momentum_ps1",
                                  "stale2_momentum"])
def test_sharded_ps_on_the_native_communicator(tmp_path, case, world):
    """The same cases on direct RCCL calls (native/rccl_comm.cpp): ncclUniqueId through the TCPStore on rank 0's
    address, ncclCommInitRank per rank, every push / pull / norm all-reduce one C call on the comm stream."""
    _need(world)
    from test_distributed import single_process_reference

    res = run_gpu_ranks(world, case, tmp_path, backend="rccl-native")
    if case.startswith("stale"):
        assert abs(res["losses"][0] - res["losses"][2]) < 1e-6 and res["losses"][-1] < res["losses"][0]
        return
    ref, _ = single_process_reference("adam" if "adam" in case else "momentum", "clip" in case)
    for n, p in ref.items():
        torch.testing.assert_close(res["params"][n], p, rtol=5e-4, atol=5e-5, msg=lambda m: "%s: %s" % (n, m))


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("case", ["real_resnet50", "real_resnet50_a2a", "real_bert", "real_bert_a2a", "real_bert_ps1"])
def test_real_layouts_bf16_wire_on_gpus(tmp_path, case, world):
    """The shipped ResNet-50 / BERT-base layouts (32 MB buckets + tail bucket, BERT's clipped AdamW with its no-decay
    group) with the bf16 wire, pipeline launched from the autograd hooks: the worker asserts after every step that
    all replicas hold bitwise the same parameters and that the master shards round to them."""
    _need(world)
    res = run_gpu_ranks(world, case, tmp_path)
    assert res["buckets"] >= 3, res
    if "bert" in case:
        assert res["groups"] == [0, 1], res
    assert all(l == l and l < 1e4 for l in res["losses"])


@pytest.mark.parametrize("world", WORLDS)
def test_stale2_on_gpus(tmp_path, world):
    _need(world)
    res = run_gpu_ranks(world, "stale2_momentum", tmp_path)
    assert abs(res["losses"][0] - res["losses"][1]) < 1e-6 and abs(res["losses"][0] - res["losses"][2]) < 1e-6
    assert abs(res["losses"][3] - res["losses"][0]) > 1e-4
    assert res["losses"][-1] < res["losses"][0]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_one_sided_async_ps_across_gpus(tmp_path, world):
    """AsyncPS with worker r on GPU r: every worker runs K1 against the other owners' shards through HIP IPC
    mappings, i.e. read-modify-write over xGMI.  Workers take turns, so the result equals sequential SGD."""
    _need(world)
    from test_distributed import _sequential_reference, run_ranks

    res = run_ranks(world, "asyncserial_momentum_multigpu", tmp_path, timeout=300)
    ref = _sequential_reference(world, "momentum")
    assert res["applies"] == 3 * world
    for n, p in ref.items():
        torch.testing.assert_close(res["params"][n], p, rtol=1e-5, atol=1e-6, msg=lambda m: "%s: %s" % (n, m))


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("case", ["widedeep", "widedeep_div", "widedeep_exact", "widedeep_capacity"])
def test_sharded_embedding_on_gpus_equals_one_gpu(tmp_path, case, world):
    """Device-side bucket-by-owner + three RCCL all-to-alls per step (ids, rows, row gradients) over W GPUs == the
    same model on one GPU with the whole batch (bf16 tower: tolerance of the bf16 GEMMs' batch split)."""
    _need(world)
    one = run_gpu_ranks(1, "widedeep", tmp_path)
    many = run_gpu_ranks(world, case, tmp_path)
    assert many["dropped"] == 0
    torch.testing.assert_close(many["table"], one["table"], rtol=5e-2, atol=2e-3)
    torch.testing.assert_close(many["acc"], one["acc"], rtol=5e-2, atol=2e-3)
    assert one["losses"][-1] < one["losses"][0] and many["losses"][-1] < many["losses"][0]


def test_sharded_embedding_single_gpu_live_group(tmp_path):
    """The W = 1 leg of the test above on its own (runs on a one-GPU box): all four routes through a live RCCL group
    leave the same table."""
    _need(1)
    ref = run_gpu_ranks(1, "widedeep", tmp_path)
    for case in ("widedeep_exact", "widedeep_capacity", "widedeep_div"):
        res = run_gpu_ranks(1, case, tmp_path)
        assert res["dropped"] == 0
        torch.testing.assert_close(res["table"], ref["table"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("case", ["embgrow_capacity", "embgrow_auto"])
def test_embedding_overflow_poll_on_gpus(tmp_path, case, world):
    """the overflow poll is a collective; the capacity factor stays identical on every rank (asserted in
    the workers after each lookup), so the equal-split all-to-alls never see mismatched sizes."""
    _need(world)
    res = run_gpu_ranks(world, case, tmp_path)
    if world == 1:
        assert res["dropped"] == 0          # one owner: cap = n, nothing can overflow
        return
    if "auto" in case:
        assert res["dropped"] == 0 and res["exact_after"] == 0, res
    else:
        assert res["dropped"] > 0 and res["factors"][-1] > res["factors"][0], res
        assert res["exact_after"] is not None, res


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("case", ["graph_mnist_mlp", "graph_bert"])
def test_whole_step_hipgraph_on_gpus(tmp_path, case, world):
    """Forward + backward + the hook-launched PS pipeline with its RCCL collectives captured in ONE hipGraph and
    replayed under a W-rank group: bitwise equal to the eager step on every rank, replicas bitwise in sync."""
    _need(world)
    res = run_gpu_ranks(world, case, tmp_path)
    assert res["buckets"] >= 2 and res["losses"][-1] < res["losses"][0]


@pytest.mark.parametrize("gpus", [2, 4, 8])
@pytest.mark.parametrize("extra", [[], ["--push", "all_to_all"], ["--graph", "1"], ["--update-mode", "stale"]])
def test_bench_under_torchrun_on_gpus(tmp_path, gpus, extra):
    """bench.py exactly as the driver launches it at N > 1 (small batch so the test is quick): one JSON line, whole-job
    images/s, the requested push / graph / update mode in ``config``."""
    _need(gpus)
    import json

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(os.path.dirname(HERE), "bench.py"),
           "--gpus", str(gpus), "--steps", "4", "--warmup", "4", "--batch", "32"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["value"] > 0 and out["config"]["global_batch"] == 32 * gpus
    if "--graph" in extra:
        assert out["config"]["hipgraph"] is True, r.stderr[-2000:]
    if "--push" in extra:
        assert out["config"]["push"] == "all_to_all"
