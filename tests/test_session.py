"""The TF1-shaped surface for user scripts (``train/session.py``): ClusterSpec / Server / replica_device_setter /
SyncReplicasOptimizer / MonitoredTrainingSession / StopAtStepHook -- the five names a script written for the reference's
image uses -- and ``examples/mnist_replica.py`` run by hand (two worker processes, then a resumed run with
ONE worker from the checkpoint the two wrote) and as the ``file`` of a REST job."""
import json
import os
import socket
import subprocess
import sys
import time
import urllib.request

import pytest
import torch

from distributed_tensorflow_amd.models import mnist_mlp
from distributed_tensorflow_amd.parallel import checkpoint as ckpt
from distributed_tensorflow_amd.parallel.comm import Comm
from distributed_tensorflow_amd.train import session as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "examples", "mnist_replica.py")


def _port():
    from conftest import free_port          # outside the kernel's ephemeral range, see there

    return free_port()


def test_cluster_spec_and_placement():
    cs = S.ClusterSpec.from_lists("a:2222", "b:2222, c:2222,")
    assert cs.jobs == ["ps", "worker"] and cs.num_tasks("worker") == 2 and cs.task_address("worker", 1) == "c:2222"
    assert S.ClusterSpec(cs.as_dict()).as_dict() == {"ps": ["a:2222"], "worker": ["b:2222", "c:2222"]}
    assert S.ClusterSpec({"worker": {0: "x:1", 1: "y:2"}}).job_tasks("worker") == ["x:1", "y:2"]
    cs2, job, idx = S.ClusterSpec.from_tf_config('{"cluster": {"ps": ["a:2222"], "worker": ["b:2222", "c:2222"]}, '
                                                 '"task": {"type": "worker", "index": 1}}')
    assert cs2.as_dict() == cs.as_dict() and (job, idx) == ("worker", 1)
    assert S.ClusterSpec.from_tf_config("{}")[1:] == ("worker", 0)
    assert S.replica_device_setter(cluster=cs).num_shards == 1
    assert S.replica_device_setter(ps_tasks=4).num_shards == 4
    assert S.replica_device_setter().num_shards is None
    with pytest.raises(ValueError, match="backup workers"):
        S.SyncReplicasOptimizer("sgd", replicas_to_aggregate=3, total_num_replicas=4)
    with pytest.raises(ValueError):
        S.StopAtStepHook()


def test_session_trains_checkpoints_and_restores(tmp_path):
    """One process: the session equals the hand-written loop on ShardedPS step for step, saves every N steps and on
    exit, and a second session over the same directory continues from the saved global_step with the saved variables."""
    from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS

    def batches():
        return [mnist_mlp.synthetic_batch(32, device="cpu", seed=k) for k in range(4)]

    torch.manual_seed(0)
    ref = mnist_mlp.build()
    ps = ShardedPS(list(ref.named_parameters()), Comm.single(), OptimizerSpec(kind="momentum", lr=0.05, momentum=0.9))
    ref_losses = []
    for i in range(12):
        loss = mnist_mlp.loss_fn(ref, batches()[i % 4])
        loss.backward()
        ps.step()
        ref_losses.append(float(loss))

    torch.manual_seed(0)
    model = mnist_mlp.build()
    d = str(tmp_path / "train")
    got = []
    with S.MonitoredTrainingSession(model, S.SyncReplicasOptimizer({"kind": "momentum", "lr": 0.05, "momentum": 0.9}),
                                    comm=Comm.single(), checkpoint_dir=d, save_checkpoint_steps=5,
                                    hooks=[S.StopAtStepHook(last_step=8)], tf_names=mnist_mlp.tf_variable_names(model)) as sess:
        bs = batches()
        while not sess.should_stop():
            got.append(float(sess.run(mnist_mlp.loss_fn(model, bs[sess.global_step % 4]))))
    assert got == ref_losses[:8]
    assert ckpt.checkpoint_steps(d) == [5, 8] and ckpt.latest_step(d) == 8

    torch.manual_seed(123)              # different initial values: the restore must replace them
    model2 = mnist_mlp.build()
    with S.MonitoredTrainingSession(model2, {"kind": "momentum", "lr": 0.05, "momentum": 0.9}, comm=Comm.single(),
                                    checkpoint_dir=d, hooks=[S.StopAtStepHook(num_steps=4)],
                                    tf_names=mnist_mlp.tf_variable_names(model2)) as sess:
        assert sess.global_step == 8
        bs = batches()
        while not sess.should_stop():
            got.append(float(sess.run(mnist_mlp.loss_fn(model2, bs[sess.global_step % 4]))))
    assert len(got) == 12 and got == pytest.approx(ref_losses, rel=1e-6, abs=1e-7)
    assert ckpt.latest_step(d) == 12


def test_chief_must_be_worker_zero():
    with pytest.raises(ValueError, match="is_chief must be worker 0"):
        S.MonitoredTrainingSession(mnist_mlp.build(), "sgd", comm=Comm.single(), is_chief=False)


def _run_workers(n, train_dir, steps, extra=()):
    hosts = ",".join("127.0.0.1:%d" % _port() for _ in range(n))
    env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="2", DTF_NUM_GPUS="0", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, SCRIPT, "--job_name=worker", "--task_index=%d" % i, "--worker_hosts=" + hosts,
                               "--train_dir=" + train_dir, "--train_steps=%d" % steps, "--batch_size=64", *extra],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(n)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    return outs


@pytest.mark.slow
def test_mnist_replica_by_hand_two_workers_then_resumed_by_one(tmp_path):
    """The sample as somebody runs it from a shell: flags, no service, no environment -- the cluster's worker 0 address
    is the rendezvous.  Two workers train 120 steps and write the checkpoint; ONE worker then resumes it to 160."""
    d = str(tmp_path / "train")
    outs = _run_workers(2, d, 120)
    assert all("global_step 120" in o for o in outs), outs
    assert "terminate called" not in "".join(outs)
    idx = ckpt.read_index(d)
    assert idx["step"] == 120 and idx["world"] == 2 and "hid_w" in idx["variables"] and "hid_w/Momentum" in idx["variables"]
    first = float(outs[0].split("loss ")[-1].split(" ->")[0])
    last = float(outs[0].split("-> ")[-1].split(" ")[0])
    assert first > 1.5 and last < 0.3, outs[0]
    outs = _run_workers(1, d, 160)
    assert "restored" in outs[0] and "global_step 160" in outs[0], outs[0]
    resumed_first = float(outs[0].split("loss ")[-1].split(" ->")[0])
    assert resumed_first < 0.5, outs[0]          # it continued, it did not start over
    assert ckpt.latest_step(d) == 160 and ckpt.read_index(d)["world"] == 1


@pytest.mark.slow
def test_mnist_replica_as_the_file_of_a_rest_job(tmp_path):
    from distributed_tensorflow_amd.service.disTf import AllInOne
    from distributed_tensorflow_amd.utils.config import ApiConfig
    from distributed_tensorflow_amd.utils.naming import parse_task_name

    ApiConfig.override("launcher", "grace_seconds", "1")
    a = AllInOne(gpus=[], log_dir=str(tmp_path / "tasks"), extra_env={"STEPS": "60", "BATCH": "64"})
    port = a.start(port=0)
    try:
        export = str(tmp_path / "export")
        body = {"file": SCRIPT, "export": export, "detail": {"ps": 1, "worker": 2}}
        req = urllib.request.Request("http://127.0.0.1:%d/v1/train" % port, data=json.dumps(body).encode(), method="POST")
        with urllib.request.urlopen(req, timeout=30) as r:
            uid = parse_task_name(json.loads(r.read().decode())[0])[0]
        t0 = time.time()
        rec = {}
        while time.time() - t0 < 180:
            rec = json.loads(a.store.get(uid))
            if "t_cleaned" in rec:
                break
            time.sleep(0.2)
        logs = ""
        for f in sorted(os.listdir(a.cluster._log_dir)):
            logs += open(os.path.join(a.cluster._log_dir, f), errors="replace").read()
        assert rec.get("status") == "succeeded", (rec, logs[-3000:])
        assert logs.count("global_step 60") == 2, logs[-3000:]
        assert ckpt.latest_step(export) == 60 and ckpt.read_index(export)["world"] == 2
    finally:
        a.stop()


@pytest.mark.gpu
def test_mnist_replica_on_the_gpu(tmp_path):
    """The same script, one worker on the MI355X: variables, PS runtime and batches live on the device, the checkpoint
    is written from there and a second run resumes it."""
    d = str(tmp_path / "train")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, SCRIPT, "--job_name=worker", "--task_index=0", "--worker_hosts=127.0.0.1:%d" % _port(),
           "--train_dir=" + d, "--batch_size=64"]
    r = subprocess.run(cmd + ["--train_steps=120"], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "global_step 120 on cuda:0" in r.stdout, r.stdout
    assert float(r.stdout.split("-> ")[-1].split(" ")[0]) < 0.3, r.stdout
    r = subprocess.run(cmd + ["--train_steps=150"], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "restored" in r.stdout and "global_step 150 on cuda:0" in r.stdout, r.stdout
    assert ckpt.latest_step(d) == 150


def test_evaluate_tool_scores_a_checkpoint_on_held_out_data(tmp_path, capsys):
    """Train on one set of synthetic digits through the session, evaluate the export directory on ANOTHER set with
    ``tools.evaluate``: loss and accuracy of the restored model, an older step on request, a clear error without data."""
    from distributed_tensorflow_amd.tools import evaluate as ev
    from distributed_tensorflow_amd.train import data as D
    from tests.test_data import _digits

    x, y = _digits(2048, seed=3)
    hx, hy = _digits(512, seed=99)
    held = tmp_path / "held"
    held.mkdir()
    D.write_idx(str(held / "t10k-images-idx3-ubyte.gz"), hx)
    D.write_idx(str(held / "t10k-labels-idx1-ubyte.gz"), hy)
    ds = D.ArrayDataset([x, y], kind="mnist")
    tf = D.transform_for(mnist_mlp, ds, "cpu")
    stream = D.BatchStream(ds, 64, seed=0)
    torch.manual_seed(0)
    model = mnist_mlp.build()
    d = str(tmp_path / "export")
    with S.MonitoredTrainingSession(model, {"kind": "momentum", "lr": 0.05, "momentum": 0.9}, comm=Comm.single(),
                                    checkpoint_dir=d, save_checkpoint_steps=10, hooks=[S.StopAtStepHook(last_step=150)],
                                    tf_names=mnist_mlp.tf_variable_names(model), keep_checkpoints=0) as sess:
        while not sess.should_stop():
            sess.run(mnist_mlp.loss_fn(model, tf([torch.from_numpy(a) for a in next(stream)])))
    assert ev.main(["--model", "mnist_mlp", "--export", d, "--data", str(held), "--batch", "128", "--device", "cpu"]) == 0
    res = json.loads(capsys.readouterr().out)
    assert res["step"] == 150 and res["samples"] == 512 and res["accuracy"] > 0.9 and res["loss"] < 0.5, res
    assert ev.main(["--model", "mnist_mlp", "--export", d, "--data", str(held), "--step", "10", "--device", "cpu"]) == 0
    early = json.loads(capsys.readouterr().out)
    assert early["step"] == 10 and early["loss"] > 10 * res["loss"] and early["accuracy"] <= res["accuracy"], (early, res)
    assert ev.main(["--model", "mnist_mlp", "--export", d, "--data", str(tmp_path / "nothing"), "--device", "cpu"]) == 1
    assert "no dataset under" in capsys.readouterr().err


@pytest.mark.gpu
def test_evaluate_restores_resnet50_on_the_gpu_exactly(tmp_path):
    """ResNet-50 (channels-last, bf16, own kernels) trained a few steps on the device, checkpointed, then scored by
    ``tools.evaluate`` in eval mode on an images.npy directory: the loss equals what the LIVE model gives on the same
    rows -- every variable and every BatchNorm moving statistic came back."""
    import types

    import numpy as np

    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS
    from distributed_tensorflow_amd.tools import evaluate as ev
    from distributed_tensorflow_amd.train import data as D
    from distributed_tensorflow_amd.train.trainer import prepare_model, save_checkpoint

    dev = torch.device("cuda", 0)
    mod = get_model_module("resnet50")
    torch.manual_seed(0)
    model = prepare_model(mod, types.SimpleNamespace(extra={}), dev, None)
    ps = ShardedPS(list(model.named_parameters()), Comm(0, 1, dev), mod.default_optimizer())
    batch = mod.synthetic_batch(32, device=dev, seed=5, image_size=96)
    for _ in range(6):
        mod.loss_fn(model, batch).backward()
        ps.step()
    d = str(tmp_path / "export")
    save_checkpoint(model, ps, d, ps.global_step, mod.tf_variable_names(model), True)
    rng = np.random.default_rng(0)
    data = tmp_path / "val"
    data.mkdir()
    img = rng.integers(0, 256, (64, 96, 96, 3), dtype=np.uint8)
    lab = rng.integers(0, 1000, 64).astype(np.int64)
    np.save(str(data / "images.npy"), img)
    np.save(str(data / "labels.npy"), lab)
    res = ev.evaluate("resnet50", d, str(data), batch=32, device="cuda:0")
    assert res["step"] == 6 and res["samples"] == 64 and 0.0 <= res["accuracy"] <= 1.0
    ds = D.open_dataset(str(data))
    tf = D.transform_for(mod, ds, dev)
    model.eval()
    want = 0.0
    with torch.no_grad():
        for lo in (0, 32):
            want += float(mod.loss_fn(model, tf([torch.from_numpy(img[lo:lo + 32]).to(dev), torch.from_numpy(lab[lo:lo + 32]).to(dev)])).double())
    # (six steps on one batch leave the moving statistics far from the batch's: the eval-mode loss is huge, which makes it
    # a sensitive probe -- one wrong statistic or variable moves it by orders of magnitude; the tolerance covers the
    # run-to-run rounding of small-batch forward passes in the default mode)
    assert res["loss"] == pytest.approx(want / 2, rel=1e-4), (res, want / 2)


@pytest.mark.slow
def test_tasks_of_a_script_job_get_tf_config(tmp_path):
    """Estimator-era scripts read ``TF_CONFIG`` instead of flags: every task of a job gets the document (cluster with the
    job's two host lists, its own type and index), and ``ClusterSpec.from_tf_config()`` reads it."""
    from distributed_tensorflow_amd.service.disTf import AllInOne
    from distributed_tensorflow_amd.utils.config import ApiConfig
    from distributed_tensorflow_amd.utils.naming import parse_task_name

    script = tmp_path / "dump_tf_config.py"
    script.write_text(
        "import json, os, sys\n"
        "from distributed_tensorflow_amd.train.session import ClusterSpec\n"
        "cs, job, idx = ClusterSpec.from_tf_config()\n"
        "out = sys.argv[7]\n"
        "os.makedirs(out, exist_ok=True)\n"
        "json.dump({'cluster': cs.as_dict(), 'job': job, 'index': idx, 'argv': sys.argv[1:5]},\n"
        "          open(os.path.join(out, 'task-%d.json' % idx), 'w'))\n")
    ApiConfig.override("launcher", "grace_seconds", "1")
    a = AllInOne(gpus=[], log_dir=str(tmp_path / "tasks"))
    port = a.start(port=0)
    try:
        export = str(tmp_path / "export")
        body = {"file": str(script), "export": export, "detail": {"ps": 1, "worker": 2}}
        req = urllib.request.Request("http://127.0.0.1:%d/v1/train" % port, data=json.dumps(body).encode(), method="POST")
        with urllib.request.urlopen(req, timeout=30) as r:
            names = json.loads(r.read().decode())
        uid = parse_task_name(names[0])[0]
        t0 = time.time()
        rec = {}
        while time.time() - t0 < 120:
            rec = json.loads(a.store.get(uid))
            if "t_cleaned" in rec:
                break
            time.sleep(0.2)
        assert rec.get("status") == "succeeded", rec
        for i in (0, 1):
            t = json.load(open(os.path.join(export, "task-%d.json" % i)))
            assert t["job"] == "worker" and t["index"] == i == int(t["argv"][1])
            assert [h.split(":")[0] for h in t["cluster"]["worker"]] == names[1:]
            assert [h.split(":")[0] for h in t["cluster"]["ps"]] == names[:1]
            assert ",".join(t["cluster"]["worker"]) == t["argv"][3]
    finally:
        a.stop()
