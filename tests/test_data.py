"""The input pipeline of the built-in workloads (train/data.py): the job's ``data`` field, which the reference hands to the
training script as argv[6] (/root/reference/handlers/train_handler.py:82-83; its sample job reads MNIST IDX files there)."""
import json
import os
import time
import urllib.request

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd.train import data as D


def _digits(n, seed=0):
    """Digits with the structure of models/mnist_mlp.synthetic_batch (class k lights up band k) as uint8 images."""
    rng = np.random.default_rng(seed)
    y = rng.integers(0, 10, n).astype(np.uint8)
    x = (rng.random((n, 784)) * 80).astype(np.uint8)
    band = 784 // 10
    for k in range(10):
        x[y == k, k * band:(k + 1) * band] += 150
    return x.reshape(n, 28, 28), y


def test_idx_round_trip_plain_and_gzipped(tmp_path):
    x, y = _digits(50)
    for name, a in (("a-idx3-ubyte", x), ("b-idx1-ubyte.gz", y)):
        p = str(tmp_path / name)
        D.write_idx(p, a)
        b = D.read_idx(p)
        assert b.dtype == np.uint8 and b.shape == a.shape and (b == a).all()
    (tmp_path / "junk").write_bytes(b"\x01\x02\x03\x04rest")
    with pytest.raises(ValueError):
        D.read_idx(str(tmp_path / "junk"))
    D.write_idx(str(tmp_path / "short"), x)
    raw = (tmp_path / "short").read_bytes()
    (tmp_path / "short").write_bytes(raw[:len(raw) // 2])
    with pytest.raises(ValueError):
        D.read_idx(str(tmp_path / "short"))


def test_open_dataset_recognises_the_three_layouts(tmp_path):
    assert D.open_dataset("") is None and D.open_dataset(str(tmp_path / "missing")) is None
    empty = tmp_path / "empty"
    empty.mkdir()
    assert D.open_dataset(str(empty)) is None             # e.g. the shipped post-data's /data/imagenet-synthetic
    x, y = _digits(40)
    m = tmp_path / "mnist"
    m.mkdir()
    D.write_idx(str(m / "train-images-idx3-ubyte.gz"), x)
    D.write_idx(str(m / "train-labels-idx1-ubyte"), y)
    ds = D.open_dataset(str(m))
    assert ds.kind == "mnist" and len(ds) == 40 and (ds.fields[0] == x).all()
    n = tmp_path / "npy"
    n.mkdir()
    np.save(n / "images.npy", np.zeros((12, 8, 8, 3), np.uint8))
    np.save(n / "labels.npy", np.arange(12))
    ds = D.open_dataset(str(n))
    assert ds.kind == "images" and len(ds) == 12 and isinstance(ds.fields[0], np.memmap)
    g = tmp_path / "generic"
    g.mkdir()
    np.save(g / "ids.npy", np.arange(30).reshape(10, 3))
    np.save(g / "dense.npy", np.ones((10, 4), np.float32))
    np.save(g / "y.npy", np.zeros(10, np.float32))
    (g / "manifest.json").write_text(json.dumps({"fields": ["ids.npy", "../generic/dense.npy", "y.npy"]}))
    ds = D.open_dataset(str(g))
    assert ds.kind == "generic" and [a.shape[1:] for a in ds.fields] == [(3,), (4,), ()]
    np.save(g / "y.npy", np.zeros(9, np.float32))
    with pytest.raises(ValueError):
        D.open_dataset(str(g))


def test_batch_stream_shards_like_between_graph_replication():
    """W workers draw the same permutation: their batches of one step are disjoint and together the global batch; an epoch
    shows every sample at most once; the next epoch is shuffled differently; the same seed repeats; a resumed stream
    continues with the batch the original one would have produced next."""
    N, W, B = 103, 4, 8
    ds = D.ArrayDataset([np.arange(N), np.arange(N) * 10])
    streams = [D.BatchStream(ds, B, worker=w, num_workers=W, seed=5) for w in range(W)]
    assert streams[0].per_epoch == N // (W * B) == 3
    seen = []
    for step in range(3):
        parts = [next(s) for s in streams]
        ids = np.concatenate([p[0] for p in parts])
        assert len(set(ids.tolist())) == W * B
        assert all((p[1] == p[0] * 10).all() for p in parts)            # fields stay aligned
        seen += ids.tolist()
    assert len(set(seen)) == len(seen) == 3 * W * B                      # one epoch: no sample twice
    epoch2 = np.concatenate([next(s)[0] for s in streams])
    again = D.BatchStream(ds, B, worker=0, num_workers=W, seed=5)
    first = next(again)[0]
    assert set(first.tolist()) == set(streams[0].indices(0).tolist())
    assert set(epoch2.tolist()) != set(seen[:W * B])
    resumed = D.BatchStream(ds, B, worker=2, num_workers=W, seed=5, start_step=4)
    assert (next(resumed)[0] == np.sort(streams[2].indices(4))).all()
    with pytest.raises(ValueError):
        D.BatchStream(ds, 64, worker=0, num_workers=2)


def test_prefetcher_delivers_the_stream_in_order_and_reports_errors():
    ds = D.ArrayDataset([np.arange(64, dtype=np.float32).reshape(64, 1), np.arange(64)])
    direct = D.BatchStream(ds, 8, seed=1)
    want = [next(direct) for _ in range(10)]
    pf = D.Prefetcher(D.BatchStream(ds, 8, seed=1), lambda ts: tuple(ts), "cpu", depth=3)
    got = [next(pf) for _ in range(10)]
    pf.close()
    for w, g in zip(want, got):
        assert (g[0].numpy() == w[0]).all() and (g[1].numpy() == w[1]).all()

    def broken():
        yield (np.zeros(2),)
        raise RuntimeError("disk gone")

    pf = D.Prefetcher(broken(), lambda ts: tuple(ts), "cpu")
    next(pf)                                                            # the good batch arrives ...
    with pytest.raises(RuntimeError, match="disk gone"):
        next(pf)                                                        # ... the failure when the next one is asked for
    pf.close()
    pf = D.Prefetcher(iter([(np.ones(1),)]), lambda ts: tuple(ts), "cpu")
    assert len(list(pf)) == 1                                           # a finite stream ends the iteration


def test_default_transforms():
    from distributed_tensorflow_amd.models import get_model_module

    x, y = _digits(6)
    f = D.transform_for(get_model_module("mnist_mlp"), D.ArrayDataset([x, y], kind="mnist"), "cpu")
    bx, by = f([torch.from_numpy(x), torch.from_numpy(y)])
    assert bx.shape == (6, 784) and bx.dtype == torch.float32 and float(bx.max()) <= 1.0 and by.dtype == torch.int64
    img = np.random.default_rng(0).integers(0, 256, (4, 16, 12, 3), dtype=np.uint8)
    lab = np.arange(4, dtype=np.int32)
    f = D.transform_for(get_model_module("resnet50"), D.ArrayDataset([img, lab], kind="images"), "cpu")
    bx, by = f([torch.from_numpy(img), torch.from_numpy(lab)])
    assert bx.shape == (4, 3, 16, 12) and bx.is_contiguous(memory_format=torch.channels_last) and by.dtype == torch.int64
    want = (img[2, 5, 7, 1] / 255.0 - 0.456) / 0.224
    assert abs(float(bx[2, 1, 5, 7]) - want) < 1e-5
    nchw = torch.randn(4, 3, 8, 8)
    bx, _ = f([nchw, torch.from_numpy(lab)])
    assert torch.equal(bx, nchw)                                         # float NCHW input passes through


def test_train_loop_takes_a_new_batch_every_step():
    """``synthetic_batches`` / dataset input: the loop consumes the iterator, one batch per step, starting with ``batch``."""
    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS
    from distributed_tensorflow_amd.train.trainer import train_loop

    mod = get_model_module("mnist_mlp")
    torch.manual_seed(0)
    model = mod.build()
    ps = ShardedPS(list(model.named_parameters()), None, mod.default_optimizer())
    bs = [mod.synthetic_batch(16, seed=s) for s in range(3)]
    seen = []
    orig = mod.loss_fn

    def spy(m, b):
        seen.append(int(b[1].sum()))
        return orig(m, b)

    mod.loss_fn = spy
    try:
        cyc = D.Cycle(bs)
        first = next(cyc)
        train_loop(model, mod, ps, first, 7, log_every=0, batches=cyc)
    finally:
        mod.loss_fn = orig
    assert seen == [int(bs[i % 3][1].sum()) for i in range(7)]


def test_scalar_log_appends_rows_and_thins_its_in_memory_copy(tmp_path):
    from distributed_tensorflow_amd.train.summary import ScalarLog, read_scalars

    seen = {}
    log = ScalarLog(str(tmp_path / "x"), sink=lambda **kw: seen.update(kw), limit=16)
    for i in range(1, 101):
        log.add(i * 10, loss=1.0 / i, items_per_sec=100.0)
    rows = read_scalars(str(tmp_path / "x"))
    assert len(rows) == 100 and rows[-1] == {"step": 1000, "wall": rows[-1]["wall"], "loss": 0.01, "items_per_sec": 100.0}
    pts = json.loads(seen["curve"])
    assert len(pts) <= 16 and pts[-1][0] == 1000 and pts[0][0] == 10          # first and newest survive, the middle is thinned
    assert [p[0] for p in pts] == sorted(p[0] for p in pts)
    with open(str(tmp_path / "x" / "scalars.jsonl"), "a") as f:
        f.write('{"step": 1010, "lo')                                           # a torn last line of a killed job
    assert len(read_scalars(str(tmp_path / "x"))) == 100
    assert ScalarLog("hdfs://nn/x").path is None and ScalarLog(None).add(1, loss=2.0)["loss"] == 2.0


@pytest.mark.slow
def test_rest_job_trains_mnist_from_idx_files_in_its_data_directory(tmp_path):
    """Config #1 with real input plumbing: POST a job whose ``data`` directory holds MNIST-format IDX files; two workers
    shard it, the loss falls well below a fixed-batch run's noise, and the log names the dataset."""
    from distributed_tensorflow_amd.service.disTf import AllInOne
    from distributed_tensorflow_amd.utils.config import ApiConfig
    from distributed_tensorflow_amd.utils.naming import parse_task_name

    x, y = _digits(4096, seed=3)
    data = tmp_path / "mnist"
    data.mkdir()
    D.write_idx(str(data / "train-images-idx3-ubyte.gz"), x)
    D.write_idx(str(data / "train-labels-idx1-ubyte.gz"), y)
    ApiConfig.override("launcher", "grace_seconds", "1")
    a = AllInOne(gpus=[], log_dir=str(tmp_path / "tasks"))
    port = a.start(port=0)
    try:
        body = {"file": "builtin://mnist_mlp?steps=60&batch=64&placement=colocated&optimizer=momentum&lr=0.05&log_every=20",
                "data": str(data), "export": str(tmp_path / "export"), "detail": {"ps": 1, "worker": 2}}
        req = urllib.request.Request("http://127.0.0.1:%d/v1/train" % port, data=json.dumps(body).encode(), method="POST")
        with urllib.request.urlopen(req, timeout=30) as r:
            names = json.loads(r.read().decode())
        uid = parse_task_name(names[0])[0]
        t0 = time.time()
        rec = {}
        while time.time() - t0 < 180:
            rec = json.loads(a.store.get(uid))
            if "t_cleaned" in rec:
                break
            time.sleep(0.2)
        logs = ""
        d = a.cluster._log_dir
        for f in sorted(os.listdir(d)):
            logs += open(os.path.join(d, f), errors="replace").read()
        assert rec.get("status") == "succeeded", (rec, logs[-3000:])
        assert "input: 4096 samples from" in logs and "(mnist), 32 steps per epoch" in logs, logs[-3000:]
        m = a.store.hgetall(uid + ":metrics")
        assert float(m["first_loss"]) > 1.5 and float(m["final_loss"]) < 0.5, m
        # the training curve: in the export directory (the chief's scalars.jsonl) and in the job's status
        from distributed_tensorflow_amd.train.summary import read_scalars

        rows = read_scalars(str(tmp_path / "export"))
        assert [r["step"] for r in rows] == [20, 40, 60] and rows[0]["loss"] > rows[-1]["loss"] and rows[-1]["items_per_sec"] > 0
        with urllib.request.urlopen("http://127.0.0.1:%d/v1/train?uid=%s" % (port, uid), timeout=30) as r:
            st = json.loads(r.read().decode())
        assert st["curve"]["columns"] == ["step", "loss", "items_per_sec"] and [p[0] for p in st["curve"]["points"]] == [20, 40, 60]
        assert "curve" not in st["metrics"]
    finally:
        a.stop()


@pytest.mark.slow
def test_failed_job_names_the_exception_and_uncapturable_graph_requests_are_declined(tmp_path):
    """Two things the option matrix on the GPU box showed (scripts/gpu_rest_matrix.py odd): a job that dies of an
    exception said only "worker task ... failed" in its record -- it now carries the exception; and ``graph=1`` with
    ``update_mode=stale`` tried a capture that cannot work -- it is now declined with a line in the log."""
    from distributed_tensorflow_amd.service.disTf import AllInOne
    from distributed_tensorflow_amd.utils.config import ApiConfig
    from distributed_tensorflow_amd.utils.naming import parse_task_name

    x, y = _digits(96, seed=3)
    data = tmp_path / "mnist"
    data.mkdir()
    D.write_idx(str(data / "train-images-idx3-ubyte.gz"), x)
    D.write_idx(str(data / "train-labels-idx1-ubyte.gz"), y)
    ApiConfig.override("launcher", "grace_seconds", "1")
    a = AllInOne(gpus=[], log_dir=str(tmp_path / "tasks"))
    port = a.start(port=0)

    def run(file, **kw):
        body = {"file": file, "detail": {"ps": 1, "worker": 1}}
        body.update(kw)
        req = urllib.request.Request("http://127.0.0.1:%d/v1/train" % port, data=json.dumps(body).encode(), method="POST")
        with urllib.request.urlopen(req, timeout=30) as r:
            uid = parse_task_name(json.loads(r.read().decode())[0])[0]
        t0 = time.time()
        rec = {}
        while time.time() - t0 < 120:
            rec = json.loads(a.store.get(uid))
            if "t_cleaned" in rec:
                break
            time.sleep(0.2)
        logs = ""
        for f in sorted(os.listdir(a.cluster._log_dir)):
            if uid in f:
                logs += open(os.path.join(a.cluster._log_dir, f), errors="replace").read()
        return rec, logs

    try:
        rec, logs = run("builtin://mnist_mlp?steps=4&batch=128", data=str(data))
        assert rec.get("status") == "failed", rec
        assert "ValueError: dataset of 96 samples is smaller than one global batch" in rec.get("reason", ""), rec
        assert "worker-0" in rec["reason"], rec
        rec, logs = run("builtin://mnist_mlp?steps=8&batch=32&update_mode=stale&staleness=3&graph=1&stpes=80")
        assert rec.get("status") == "succeeded", (rec, logs[-2000:])
        assert "graph=1 ignored: a step with update_mode=stale cannot be captured" in logs, logs[-2000:]
        assert "WARNING: option(s) stpes are not read by anything" in logs, logs[-2000:]     # a typo does not pass silently
    finally:
        a.stop()


@pytest.mark.gpu
def test_prefetcher_on_the_gpu_matches_the_host_transform():
    """uint8 NHWC images through the pinned-host -> copy-stream -> normalise path while the training stream is busy: every
    batch equals the CPU transform of the same rows (bf16, channels-last), labels int64."""
    from distributed_tensorflow_amd.models import get_model_module

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (256, 24, 20, 3), dtype=np.uint8)
    lab = rng.integers(0, 1000, 256).astype(np.int32)
    ds = D.ArrayDataset([img, lab], kind="images")
    mod = get_model_module("resnet50")
    pf = D.Prefetcher(D.BatchStream(ds, 32, seed=2), D.transform_for(mod, ds, dev), dev)
    ref_stream = D.BatchStream(ds, 32, seed=2)
    cpu_f = D.transform_for(mod, ds, "cpu")
    busy = torch.randn(4096, 4096, device=dev)
    for _ in range(12):
        busy = busy @ busy * 1e-4                       # keep the compute stream occupied between batches
        bx, by = next(pf)
        rx, ry = cpu_f([torch.from_numpy(a) for a in next(ref_stream)])
        assert bx.dtype == torch.bfloat16 and bx.is_contiguous(memory_format=torch.channels_last) and by.dtype == torch.int64
        torch.testing.assert_close(bx.float().cpu(), rx.to(torch.bfloat16).float(), rtol=0, atol=0)
        assert torch.equal(by.cpu(), ry)
    pf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_training_from_a_dataset_on_the_gpu(graph, tmp_path):
    """mnist_mlp on the GPU fed by the prefetcher from IDX files, eagerly and with the step captured in a hipGraph (new
    batches are copied into the captured input tensors): the loss falls on data that changes every step."""
    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS
    from distributed_tensorflow_amd.train.trainer import train_loop

    dev = torch.device("cuda", 0)
    x, y = _digits(2048, seed=4)
    D.write_idx(str(tmp_path / "train-images-idx3-ubyte"), x)
    D.write_idx(str(tmp_path / "train-labels-idx1-ubyte"), y)
    ds = D.open_dataset(str(tmp_path))
    mod = get_model_module("mnist_mlp")
    torch.manual_seed(0)
    model = mod.build().to(dev)
    ps = ShardedPS(list(model.named_parameters()), None, OptimizerSpec(kind="momentum", lr=0.05, momentum=0.9))
    pf = D.Prefetcher(D.BatchStream(ds, 64, seed=0), D.transform_for(mod, ds, dev), dev)
    first = next(pf)
    losses = []
    train_loop(model, mod, ps, first, 40, log_every=0, graph=graph, batches=pf,
               on_step=lambda step, loss: losses.append(loss.detach().float().clone()))
    pf.close()
    ls = [float(v) for v in losses]
    assert ls[0] > 1.5 and sum(ls[-5:]) / 5 < 0.5, ls


@pytest.mark.slow
def test_rest_job_runs_a_user_script_with_the_reference_arguments(tmp_path):
    """The reference's mode of use: ``file`` is a python script, run in every worker task with the nine task arguments
    (handlers/train_handler.py:82-83) -- here examples/train_mnist_script.py, two workers, MNIST IDX files in ``data``,
    ``export`` for its result.  The ps task is parked and reaped when the workers are done."""
    from distributed_tensorflow_amd.service.disTf import AllInOne
    from distributed_tensorflow_amd.utils.config import ApiConfig
    from distributed_tensorflow_amd.utils.naming import parse_task_name

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    x, y = _digits(2048, seed=5)
    data = tmp_path / "mnist"
    data.mkdir()
    D.write_idx(str(data / "train-images-idx3-ubyte"), x)
    D.write_idx(str(data / "train-labels-idx1-ubyte"), y)
    export = tmp_path / "export"
    ApiConfig.override("launcher", "grace_seconds", "1")
    a = AllInOne(gpus=[], log_dir=str(tmp_path / "tasks"))
    port = a.start(port=0)
    try:
        body = {"file": os.path.join(root, "examples", "train_mnist_script.py"), "data": str(data), "export": str(export),
                "detail": {"ps": 1, "worker": 2}}
        req = urllib.request.Request("http://127.0.0.1:%d/v1/train" % port, data=json.dumps(body).encode(), method="POST")
        with urllib.request.urlopen(req, timeout=30) as r:
            names = json.loads(r.read().decode())
        assert len(names) == 3
        uid = parse_task_name(names[0])[0]
        t0 = time.time()
        rec = {}
        while time.time() - t0 < 180:
            rec = json.loads(a.store.get(uid))
            if "t_cleaned" in rec:
                break
            time.sleep(0.2)
        logs = ""
        d = a.cluster._log_dir
        for f in sorted(os.listdir(d)):
            logs += open(os.path.join(d, f), errors="replace").read()
        assert rec.get("status") == "succeeded", (rec, logs[-3000:])
        res = json.load(open(export / "result.json"))
        assert res["workers"] == 2 and res["dataset"] == 2048 and res["steps"] == 60
        assert res["first_loss"] > 1.5 and res["final_loss"] < 0.5, res
        assert "[worker 0] 60 steps" in logs and "[worker 1] 60 steps" in logs, logs[-3000:]
        assert a.cluster.dns_map(uid) == {}                      # ps task reaped, services gone
    finally:
        a.stop()


@pytest.mark.gpu
@pytest.mark.parametrize("margin,expect", [(0.98, "replaying"), (0.0, "staying eager")])
def test_graph_auto_times_eager_against_replay_and_keeps_the_faster(margin, expect, capsys, monkeypatch):
    """``graph=auto``: eager steps are timed before the capture, replays after it; the tiny MLP is host-bound, so the replay
    wins -- unless the margin says it never may (0.0), which exercises the way back: graph and memory pool dropped, eager
    steps continue on the default stream.  Either way the loop trains (changing batches) and ends with the same
    parameters as the all-eager loop."""
    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS
    from distributed_tensorflow_amd.train import trainer

    monkeypatch.setattr(trainer, "GRAPH_AUTO_MARGIN", margin)
    dev = torch.device("cuda", 0)
    mod = get_model_module("mnist_mlp")
    ends = []
    for graph in (False, "auto"):
        torch.manual_seed(0)
        model = mod.build().to(dev)
        ps = ShardedPS(list(model.named_parameters()), None, OptimizerSpec(kind="momentum", lr=0.05, momentum=0.9))
        cyc = D.Cycle([mod.synthetic_batch(64, device=dev, seed=s) for s in range(5)])
        first = next(cyc)
        last, _, n = trainer.train_loop(model, mod, ps, first, 80, log_every=0, graph=graph, batches=cyc)
        assert n == 80 and last < 0.2
        ends.append(ps.master.clone())
        ps.remove_hooks()
    out = capsys.readouterr().out
    assert "hipGraph trial: eager" in out and expect in out, out
    assert torch.equal(ends[0], ends[1])


def test_learning_rate_schedule_options_and_their_use_in_the_training_loop():
    """``lr_warmup_steps`` / ``lr_schedule=cosine|piecewise`` (job options): the values, and that ``train_loop`` hands the
    rate of the JOB's step to the PS runtime -- a run resumed at step 6 continues the schedule instead of restarting it."""
    import types

    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS
    from distributed_tensorflow_amd.train.jobspec import parse_file_field
    from distributed_tensorflow_amd.train.trainer import lr_schedule, train_loop

    _, o = parse_file_field("builtin://mnist_mlp?lr_warmup_steps=4&lr_schedule=piecewise&lr_boundaries=0.5,0.75&lr_decay_factor=0.1")
    f = lr_schedule(o, 0.8, 20)
    assert [round(f(t), 6) for t in (0, 1, 3, 4, 9, 10, 14, 15, 19)] == [0.2, 0.4, 0.8, 0.8, 0.8, 0.08, 0.08, 0.008, 0.008]
    _, o = parse_file_field("builtin://mnist_mlp?lr_warmup_steps=2&lr_schedule=cosine&lr_final_fraction=0.1")
    f = lr_schedule(o, 1.0, 12)
    assert f(0) == 0.5 and f(1) == 1.0 and f(2) == 1.0 and abs(f(11) - 0.1) < 1e-12 and f(6) < f(4) < f(2)
    assert lr_schedule(parse_file_field("builtin://mnist_mlp")[1], 0.1, 10) is None
    with pytest.raises(ValueError):
        lr_schedule(types.SimpleNamespace(lr_schedule="exponential"), 0.1, 10)

    mod = get_model_module("mnist_mlp")
    seen = []

    def run(start, n):
        torch.manual_seed(0)
        model = mod.build()
        ps = ShardedPS(list(model.named_parameters()), Comm.single(), OptimizerSpec(kind="sgd", lr=1.0))
        orig = ps.step

        def step():
            seen.append(round(ps.lr, 6))
            return orig()

        ps.step = step
        train_loop(model, mod, ps, mod.synthetic_batch(16, device="cpu", seed=1), n, log_every=0, start_step=start, lr_fn=f)

    run(0, 12)
    whole = list(seen)
    del seen[:]
    run(6, 6)
    assert whole == [round(f(t), 6) for t in range(12)] and seen == whole[6:]
