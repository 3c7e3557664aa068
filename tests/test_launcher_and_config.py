"""Launcher scheduling (amd.com/gpu accounting, Pending jobs, HIP_VISIBLE_DEVICES), config
overrides and job-spec parsing."""
import json
import os
import time

import pytest

from distributed_tensorflow_amd.service import manifest
from distributed_tensorflow_amd.service.launcher import ApiException, LocalCluster
from distributed_tensorflow_amd.train.jobspec import parse_file_field
from distributed_tensorflow_amd.utils.config import ApiConfig

UID = "3e8f3702-d10b-11e8-abe4-fa163ef8da8a"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wait(cond, timeout=10.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if cond():
            return True
        time.sleep(0.05)
    return cond()


@pytest.fixture
def cluster(tmp_path):
    ApiConfig.override("image", "tensorflow", "tests.fake_task")
    ApiConfig.override("launcher", "grace_seconds", "0.3")
    ApiConfig.override("launcher", "verify_reclaim", "0")        # no real devices here
    c = LocalCluster(gpus=[0, 1], log_dir=str(tmp_path), extra_env={"PYTHONPATH": ROOT})
    yield c
    c.shutdown()
    ApiConfig.override("launcher", "verify_reclaim", "1")


@pytest.fixture
def cluster3(tmp_path):
    ApiConfig.override("image", "tensorflow", "tests.fake_task")
    ApiConfig.override("launcher", "grace_seconds", "0.3")
    ApiConfig.override("launcher", "verify_reclaim", "0")
    c = LocalCluster(gpus=[0, 1, 2], log_dir=str(tmp_path), extra_env={"PYTHONPATH": ROOT})
    yield c
    c.shutdown()
    ApiConfig.override("launcher", "verify_reclaim", "1")


@pytest.fixture
def probing_cluster(tmp_path, monkeypatch):
    """Two fake devices with the HBM verification ON, the hipMemGetInfo child replaced by a slow fake: the window
    between 'task killed' and 'devices released' is then wide enough to race against."""
    from distributed_tensorflow_amd.service import launcher as L

    calls = []

    def fake_probe(gpus, python=None, timeout=90):
        calls.append(list(gpus))
        time.sleep(0.4)
        return [{"gpu": g, "free": 1000, "total": 1100, "used": 100} for g in gpus]

    monkeypatch.setattr(L, "probe_hbm", fake_probe)
    monkeypatch.setattr(L, "hbm_used_bytes", lambda: [])
    ApiConfig.override("image", "tensorflow", "tests.fake_task")
    ApiConfig.override("launcher", "grace_seconds", "0.3")
    ApiConfig.override("launcher", "verify_reclaim", "1")
    ApiConfig.override("launcher", "hbm_sample_wait_seconds", "0.2")
    c = LocalCluster(gpus=[0, 1], log_dir=str(tmp_path), extra_env={"PYTHONPATH": ROOT})
    c.probe_calls = calls
    yield c
    c.shutdown()


def _holders(cluster):
    """device -> names of the tasks that currently hold it (running or being verified)."""
    out = {}
    with cluster._lock:
        for t in cluster._jobs.values():
            if t.gpus and not t.released:
                for g in t.gpus:
                    out.setdefault(g, []).append(t.name)
        free = list(cluster._free_gpus)
    return out, free


def test_delete_while_creating_never_double_allocates_a_gpu(probing_cluster):
    """cancel a job and resubmit at once.  The killed task's devices must stay reserved until its HBM
    probe is over and go back to the free list exactly once -- no device may ever have two tenants, nor be free
    while held."""
    import threading

    c = probing_cluster
    bad = []
    stop = threading.Event()

    def invariant():
        while not stop.is_set():
            held, free = _holders(c)
            for g, names in held.items():
                if len(names) > 1 or g in free:
                    bad.append((g, names, free))
            if len(set(free)) != len(free):
                bad.append(("dup-free", free))
            time.sleep(0.002)

    th = threading.Thread(target=invariant, daemon=True)
    th.start()
    for rnd in range(3):
        a = c.create_namespaced_job("default", _job(4 * rnd + 0, 99, 1, "sleep:30"))
        b = c.create_namespaced_job("default", _job(4 * rnd + 1, 99, 1, "sleep:30"))
        assert sorted(a["status"]["gpus"] + b["status"]["gpus"]) == [0, 1]
        deleters = [threading.Thread(target=c.delete_namespaced_job, args=(j["metadata"]["name"], "default"))
                    for j in (a, b)]
        for d in deleters:
            d.start()
        # resubmit while the kills / probes are in flight: must stay Pending until a device is REALLY back
        n1 = c.create_namespaced_job("default", _job(4 * rnd + 2, 99, 1, "sleep:0.2"))
        n2 = c.create_namespaced_job("default", _job(4 * rnd + 3, 99, 1, "sleep:0.2"))
        for d in deleters:
            d.join()
        assert _wait(lambda: all(c.read_namespaced_job(j["metadata"]["name"], "default")["status"]["succeeded"] == 1
                                 for j in (n1, n2)), 20)
        assert _wait(lambda: c.free_gpus() == [0, 1], 10)
    stop.set()
    th.join()
    assert not bad, bad[:5]
    # every ended GPU task was probed (killed ones by delete, finished ones by their waiter thread)
    reps = c.reclaim_reports
    assert len(reps) == 12 and all(r["reclaimed"] for r in reps)
    assert any("during" in r for r in reps)


def test_event_log_is_bounded(cluster):
    cluster._max_events, cluster._event_ttl = 40, 0.0
    for i in range(30):
        j = cluster.create_namespaced_job("default", _job(i, 77, 0, "exit:0"))
        name = j["metadata"]["name"]
        assert _wait(lambda: cluster.read_namespaced_job(name, "default")["status"]["succeeded"] == 1)
        cluster.delete_namespaced_job(name, "default")
    assert len(cluster._events) <= 41
    # a watcher that resumes from an old resourceVersion neither crashes nor repeats events
    evs = list(cluster.watch("default", 5, timeout=0.2))
    rvs = [int(e["object"]["metadata"]["resourceVersion"]) for e in evs]
    assert rvs == sorted(set(rvs)) and all(r > 5 for r in rvs)


def _job(seq, count, gpu, file_field="sleep:0.4"):
    info = {"file": file_field, "resource": {"gpu": gpu}, "detail": {"ps": 0, "worker": count}}
    body = manifest.gen_job(UID, "worker", seq, count, info, "", "")
    body["metadata"].pop("annotations")       # these tests schedule each task by itself (no gang, see the gang tests below)
    return body


def _gang(uid, workers, gpu=1, ps=0, ps_gpu=0, file_field="sleep:0.4"):
    info = {"file": file_field, "resource": {"gpu": gpu, "ps_gpu": ps_gpu}, "detail": {"ps": ps, "worker": workers}}
    return manifest.gen_all(uid, info)[1]


def test_tasks_of_one_job_get_their_devices_together_or_not_at_all(cluster3):
    """Gang scheduling.  Two 2-GPU jobs whose tasks arrive interleaved on a 3-GPU node: without a reservation each
    would start one task, the third device would go to whoever comes next, and the loser would wait in its rendezvous
    holding a device.  With it the first job owns two devices from its first task on, the second job's tasks (its
    device-less ps task too) stay Pending until two devices are free, then all start."""
    c = cluster3
    if True:
        ja = _gang("11111111-aaaa-11f1-8000-000000000001", 2, file_field="sleep:1.0")
        jb = _gang("22222222-bbbb-11f1-8000-000000000002", 2, ps=1, file_field="sleep:0.3")
        a0 = c.create_namespaced_job("default", ja[0])
        b_ps = c.create_namespaced_job("default", jb[0])          # job B's ps task: no device, still waits for the gang
        b0 = c.create_namespaced_job("default", jb[1])
        assert a0["status"]["gpus"] == [0] and c.free_gpus() == [2]            # device 1 is set aside for a1
        assert b_ps["status"]["phase"] == "Pending" and b0["status"]["phase"] == "Pending"
        b1 = c.create_namespaced_job("default", jb[2])
        assert b1["status"]["phase"] == "Pending" and c.free_gpus() == [2]
        a1 = c.create_namespaced_job("default", ja[1])
        assert a1["status"]["gpus"] == [1]
        names_b = [j["metadata"]["name"] for j in (b_ps, b0, b1)]
        assert _wait(lambda: all(c.read_namespaced_job(n, "default")["status"]["succeeded"] == 1 for n in names_b[1:]), 20)
        assert c.read_namespaced_job(names_b[0], "default")["status"]["phase"] == "Running"     # a ps task joins forever
        got = sorted(g for n in names_b for g in c.read_namespaced_job(n, "default")["status"]["gpus"])
        assert len(got) == 2 and len(set(got)) == 2
        for j in (a0, a1, b_ps, b0, b1):
            c.delete_namespaced_job(j["metadata"]["name"], "default")
        assert c.free_gpus() == [0, 1, 2] and not c._gangs and not c._gang_broken


def test_job_that_needs_more_devices_at_once_than_the_node_has_is_refused(cluster):
    c = cluster
    if True:
        with pytest.raises(ApiException) as e:
            c.create_namespaced_job("default", _gang("33333333-cccc-11f1-8000-000000000003", 3)[0])
        assert e.value.status == 422 and "need 3 amd.com/gpu at once, node has 2" in e.value.reason
        with pytest.raises(ApiException) as e:      # two workers and a ps task with a device of its own
            c.create_namespaced_job("default", _gang("44444444-dddd-11f1-8000-000000000004", 2, ps=1, ps_gpu=1)[0])
        assert e.value.status == 422
        assert c.free_gpus() == [0, 1]


def test_devices_set_aside_for_a_job_come_back_when_it_is_torn_down_half_created(cluster):
    c = cluster
    if True:
        j = _gang("55555555-eeee-11f1-8000-000000000005", 2, file_field="sleep:5")
        a0 = c.create_namespaced_job("default", j[0])
        assert c.free_gpus() == []                 # one in use, one set aside
        c.delete_namespaced_job(a0["metadata"]["name"], "default")         # submit rolled back / job cancelled
        assert c.free_gpus() == [0, 1] and not c._gangs and not c._gang_broken


def test_gpu_accounting_and_pending_jobs(cluster):
    a = cluster.create_namespaced_job("default", _job(0, 3, 1))
    b = cluster.create_namespaced_job("default", _job(1, 3, 1))
    c = cluster.create_namespaced_job("default", _job(2, 3, 1))      # no device left -> Pending
    assert a["status"]["gpus"] == [0] and b["status"]["gpus"] == [1]
    assert c["status"]["phase"] == "Pending" and cluster.free_gpus() == []
    # the task sees exactly its device
    name = a["metadata"]["name"]
    assert cluster._jobs[("default", name)].proc is not None
    # when a worker exits its device is released and the pending job starts
    assert _wait(lambda: cluster.read_namespaced_job(c["metadata"]["name"], "default")["status"]["phase"]
                 in ("Running", "Succeeded"))
    assert _wait(lambda: all(cluster.read_namespaced_job(j["metadata"]["name"], "default")["status"]["succeeded"] == 1
                             for j in (a, b, c)))
    assert cluster.free_gpus() == [0, 1]
    types = [e["type"] for e in cluster.watch("default", 0, timeout=0.3)]
    assert types.count("ADDED") == 3 and "MODIFIED" in types


def test_request_larger_than_node_is_rejected(cluster):
    with pytest.raises(ApiException) as e:
        cluster.create_namespaced_job("default", _job(0, 1, 4))
    assert e.value.status == 422


def test_failed_task_reports_failed_status_and_exit_code(cluster):
    j = cluster.create_namespaced_job("default", _job(0, 1, 0, "exit:7"))
    name = j["metadata"]["name"]
    assert _wait(lambda: cluster.read_namespaced_job(name, "default")["status"]["failed"] == 1)
    st = cluster.read_namespaced_job(name, "default")["status"]
    assert st["exitCode"] == 7 and st["succeeded"] is None


def test_watch_resumes_from_resource_version(cluster):
    cluster.create_namespaced_job("default", _job(0, 2, 0, "sleep:5"))
    evs = list(cluster.watch("default", 0, timeout=0.3))
    rv = int(evs[-1]["object"]["metadata"]["resourceVersion"])
    cluster.create_namespaced_job("default", _job(1, 2, 0, "sleep:5"))
    later = list(cluster.watch("default", rv, timeout=0.3))
    assert later and all(int(e["object"]["metadata"]["resourceVersion"]) > rv for e in later)
    assert later[0]["type"] == "ADDED" and later[0]["object"]["metadata"]["name"].endswith("-worker-1-2")


def test_duplicate_names_conflict(cluster):
    cluster.create_namespaced_service("default", manifest.gen_service(UID, "ps", 0, 1))
    with pytest.raises(ApiException) as e:
        cluster.create_namespaced_service("default", manifest.gen_service(UID, "ps", 0, 1))
    assert e.value.status == 409
    host, port = cluster.resolve("tf-%s-ps-0-1" % UID).split(":")
    assert host == "127.0.0.1" and int(port) > 0


def test_config_sections_and_env_override(monkeypatch):
    cfg = ApiConfig()
    # the reference's keys (conf/config:1-26) ...
    assert cfg.getint("service", "port") == 12345
    assert cfg.get("namespace", "tensorflow") == "default"
    assert cfg.getint("k8s", "headless_port") == 2222
    assert cfg.get("redis", "running_set") == "running_uids" and cfg.get("redis", "success_set") == "success_uids"
    # ... the section it forgot ...
    for k in ("handlers", "ps_key", "worker_key", "delete_queue"):
        assert cfg.get("event", k)
    # ... and the new ones
    assert cfg.get("launcher", "gpu_resource") == "amd.com/gpu"
    monkeypatch.setenv("DTF_SERVICE_PORT", "0")
    assert ApiConfig().getint("service", "port") == 0
    assert ApiConfig() is cfg


def test_jobspec_file_field():
    kind, o = parse_file_field("builtin://resnet50?steps=7&batch=32&update_mode=stale&fault=exit:worker=1,step=3&depth=101")
    assert kind == "builtin" and o.model == "resnet50" and o.steps == 7 and o.batch == 32
    assert o.update_mode == "stale" and o.fault.startswith("exit:") and o.extra == {"depth": "101"}
    assert parse_file_field("inception-v3")[1].model == "inception_v3"
    kind, path = parse_file_field(__file__)
    assert kind == "script" and path == __file__
    with pytest.raises(ValueError):
        parse_file_field("/xxx/")          # the reference's sample value is neither


def test_warm_rank_pool_adopts_a_task(tmp_path):
    """[launcher] warm_pool: a process that has already imported torch + the package becomes the task (one JSON line
    on its stdin: argv, environment delta, log file); exit code, log and lifecycle events are those of a cold task."""
    ApiConfig.override("launcher", "warm_pool", "2")
    ApiConfig.override("launcher", "verify_reclaim", "0")
    c = LocalCluster(gpus=[], log_dir=str(tmp_path), extra_env={"PYTHONPATH": ROOT, "DTF_REDIS_PORT": "1"})
    try:
        assert c._warm is not None
        assert _wait(lambda: c._warm.ready() == 2, 120), "warm workers did not come up"
        info = {"file": "builtin://mnist_mlp?steps=3&batch=16&log_every=1", "resource": {"gpu": 0},
                "detail": {"ps": 0, "worker": 1}}
        c.create_namespaced_service("default", manifest.gen_service(UID, "worker", 0, 1))
        t0 = time.time()
        j = c.create_namespaced_job("default", manifest.gen_job(UID, "worker", 0, 1, info, "",
                                                                "tf-%s-worker-0-1:2222" % UID))
        name = j["metadata"]["name"]
        assert j["status"].get("warm") is True
        assert _wait(lambda: c.read_namespaced_job(name, "default")["status"]["succeeded"] == 1, 120), \
            open(os.path.join(str(tmp_path), name + ".log")).read()[-2000:]
        took = time.time() - t0
        text = open(os.path.join(str(tmp_path), name + ".log")).read()
        assert "[warm] pid" in text and "[train] step 3" in text
        assert _wait(lambda: c._warm.ready() == 2, 120), "the pool must refill"
        print("warm task wall time %.2f s" % took)
    finally:
        c.shutdown()


def test_placement_is_resolved_per_job_not_per_task():
    """with placement=auto a GPU-less ps task used to choose 'dedicated' while its GPU workers chose
    'colocated'.  Every task of a job now derives the placement from the job-level facts in its environment."""
    from distributed_tensorflow_amd.service.entry import resolve_placement

    info = {"file": "resnet50", "resource": {"gpu": 1}, "detail": {"ps": 1, "worker": 2}}
    jobs = [manifest.gen_job(UID, t, 0, n, info, "a", "b") for t, n in (("ps", 1), ("worker", 2))]
    envs = []
    for j, own_gpus in zip(jobs, ("0", "1")):          # the ps task has no device of its own
        env = {e["name"]: e["value"] for e in j["spec"]["template"]["spec"]["containers"][0]["env"]}
        env["DTF_NUM_GPUS"] = own_gpus
        envs.append(env)
    assert [resolve_placement("auto", 1, e) for e in envs] == ["colocated", "colocated"]
    # ps tasks with their own GPUs, or a CPU job with ps tasks: dedicated -- again for both task kinds
    assert resolve_placement("auto", 2, {"DTF_JOB_WORKER_GPUS": "1", "DTF_JOB_PS_GPUS": "1", "DTF_NUM_GPUS": "0"}) == "dedicated"
    assert resolve_placement("auto", 1, {"DTF_JOB_WORKER_GPUS": "0", "DTF_JOB_PS_GPUS": "0"}) == "dedicated"
    assert resolve_placement("auto", 0, {"DTF_JOB_WORKER_GPUS": "0"}) == "colocated"
    # an explicit request stands -- except where the group could never form: GPU workers (RCCL) with device-less ps
    # tasks (gloo only); both task kinds then take the colocated way (the job used to hang in its rendezvous)
    assert resolve_placement("dedicated", 2, {"DTF_JOB_WORKER_GPUS": "1", "DTF_JOB_PS_GPUS": "1"}) == "dedicated"
    assert resolve_placement("dedicated", 1, {"DTF_JOB_WORKER_GPUS": "0", "DTF_JOB_PS_GPUS": "0"}) == "dedicated"
    assert [resolve_placement("dedicated", 1, e) for e in envs] == ["colocated", "colocated"]
    assert resolve_placement("colocated", 1, envs[1]) == "colocated"


def test_miopen_setup_restricts_solver_families(tmp_path, monkeypatch):
    """a shape missing from the shipped MIOpen caches must not cost a 36 s JIT stall.  The
    non-implicit-GEMM solver families (the ones that compile HIP / OpenCL at run time) are switched off unless the
    user exported a value or asks for MIOpen's defaults."""
    from distributed_tensorflow_amd.utils import miopen

    for k in miopen._NON_IGEMM + ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR", "DTF_MIOPEN_ALL_SOLVERS"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.setenv("MIOPEN_DEBUG_CONV_WINOGRAD", "1")          # an explicit user choice wins
    db = miopen.setup()
    assert db and os.path.isdir(db) and any(f.endswith(".txt") for f in os.listdir(db))
    assert os.environ["MIOPEN_DEBUG_CONV_DIRECT"] == "0" and os.environ["MIOPEN_DEBUG_CONV_GEMM"] == "0"
    assert os.environ["MIOPEN_DEBUG_CONV_FFT"] == "0" and os.environ["MIOPEN_DEBUG_CONV_WINOGRAD"] == "1"
    cache = os.environ["MIOPEN_CUSTOM_CACHE_DIR"]
    assert any(f.endswith(".ukdb") for f in os.listdir(cache))
    for k in miopen._NON_IGEMM:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DTF_MIOPEN_ALL_SOLVERS", "1")
    assert miopen.restrict_solvers() is False and "MIOPEN_DEBUG_CONV_DIRECT" not in os.environ


def test_gemm_tuning_setup_is_read_only_and_respects_the_user(tmp_path, monkeypatch):
    from distributed_tensorflow_amd.utils import gemm_tuning

    for k in list(os.environ):
        if k.startswith("PYTORCH_TUNABLEOP_") or k == "DTF_GEMM_TUNING":
            monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    f = gemm_tuning.setup()
    assert f and os.environ["PYTORCH_TUNABLEOP_ENABLED"] == "1" and os.environ["PYTORCH_TUNABLEOP_TUNING"] == "0"
    # one copy per device ordinal (TunableOp inserts the ordinal before the extension), validators first
    d = os.path.dirname(f)
    assert os.path.basename(d) == str(os.getpid())          # private per process
    assert all(os.path.isfile(os.path.join(d, "gemm%d.csv" % i)) for i in range(8))
    lines = open(os.path.join(d, "gemm3.csv")).read().splitlines()
    assert lines[0].startswith("Validator,") and any(l.startswith("GemmTunableOp_BFloat16") for l in lines)
    # a user who drives TunableOp himself is left alone; DTF_GEMM_TUNING=0 switches the feature off
    monkeypatch.setenv("PYTORCH_TUNABLEOP_TUNING", "1")
    monkeypatch.setenv("PYTORCH_TUNABLEOP_FILENAME", "mine.csv")
    assert gemm_tuning.setup() == "mine.csv" and os.environ["PYTORCH_TUNABLEOP_TUNING"] == "1"
    for k in list(os.environ):
        if k.startswith("PYTORCH_TUNABLEOP_"):
            monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DTF_GEMM_TUNING", "0")
    assert gemm_tuning.setup() is None and "PYTORCH_TUNABLEOP_ENABLED" not in os.environ


def test_tuned_gemm_shapes_drive_the_conv1x1_library_choice(tmp_path, monkeypatch):
    """A 1x1 convolution product below the channel threshold goes to the GEMM library only when the shipped solution
    choices are active and hold that exact GEMM."""
    import torch

    from distributed_tensorflow_amd.ops import conv
    from distributed_tensorflow_amd.utils import gemm_tuning

    for k in list(os.environ):
        if k.startswith("PYTORCH_TUNABLEOP_") or k == "DTF_GEMM_TUNING":
            monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.setattr(gemm_tuning, "_active", False)
    monkeypatch.setattr(gemm_tuning, "_fwd_shapes", None)
    x = torch.empty(256, 64, 56, 56, device="meta")
    w_expand = torch.empty(256, 64, 1, 1, device="meta")
    w_256 = torch.empty(64, 256, 1, 1, device="meta")
    assert gemm_tuning.tuned_gemm_shapes() == frozenset()
    assert not conv._mm_fwd(x, w_expand)                                    # default heuristics: MIOpen
    assert conv._mm_fwd(torch.empty(8, 1024, 14, 14, device="meta"), torch.empty(256, 1024, 1, 1, device="meta"))
    assert gemm_tuning.setup()
    shapes = gemm_tuning.tuned_gemm_shapes()
    assert ("tn", 256, 256 * 56 * 56, 64) in shapes                          # ResNet-50 stage 1 expand, batch 256
    assert conv._mm_fwd(x, w_expand)
    assert not conv._mm_fwd(torch.empty(64, 64, 56, 56, device="meta"), w_expand)       # another batch size: not recorded
    assert not conv._mm_fwd(torch.empty(256, 256, 56, 56, device="meta"), w_256)        # 256 input channels: MIOpen measured faster
    monkeypatch.setenv("PYTORCH_TUNABLEOP_ENABLED", "0")
    assert gemm_tuning.tuned_gemm_shapes() == frozenset()


def test_gpu_hot_pool_hands_the_device_its_own_rank_and_respawns(tmp_path):
    """GPU-hot ranks: one pre-started rank per free device, pinned with HIP_VISIBLE_DEVICES; a
    1-GPU task adopts the rank of the device it was scheduled on, nobody else stays on that device, and the device
    gets a fresh rank once it is released.  (No device here: the rank skips the HIP part, the plumbing is the same.)"""
    from distributed_tensorflow_amd.service.launcher import ENTRY_IMAGE, LocalCluster
    from distributed_tensorflow_amd.service.manifest import gen_job

    ApiConfig.override("image", "tensorflow", ENTRY_IMAGE)
    ApiConfig.override("launcher", "verify_reclaim", "0")
    ApiConfig.override("launcher", "warm_gpu_pool", "1")
    try:
        c = LocalCluster(gpus=[0, 1], log_dir=str(tmp_path), extra_env={"PYTHONPATH": ROOT})
        assert c._hot is not None
        assert _wait(lambda: c._hot.ready() == [0, 1], 120), c._hot.ready()
        uid = "3e8f3702-d10b-11e8-abe4-fa163ef8da8a"
        info = {"file": "builtin://mnist_mlp?steps=3&batch=8&log_every=1", "resource": {"gpu": 1},
                "detail": {"ps": 0, "worker": 1}}
        body = gen_job(uid, "worker", 0, 1, info, "", "tf-%s-worker-0-1:2222" % uid)
        j = c.create_namespaced_job("default", body)
        name = j["metadata"]["name"]
        st = c.read_namespaced_job(name, "default")["status"]
        assert st["warm"] == "hot" and st["gpus"] == [0]
        assert 0 not in c._hot.ready()                      # the device's rank became the task
        assert _wait(lambda: c.read_namespaced_job(name, "default")["status"]["succeeded"] == 1, 120), \
            open(os.path.join(str(tmp_path), name + ".log")).read()
        log_text = open(os.path.join(str(tmp_path), name + ".log")).read()
        assert "GPU-hot" in log_text and "done: 3 steps" in log_text
        assert _wait(lambda: c._hot.ready() == [0, 1], 120)   # respawned after the release
    finally:
        c.shutdown()
        ApiConfig.override("launcher", "warm_gpu_pool", "0")
        ApiConfig.override("launcher", "verify_reclaim", "1")
        ApiConfig.override("image", "tensorflow", "tests.fake_task")


def test_miopen_solver_restriction_is_undone_for_user_scripts(monkeypatch):
    """``restrict_solvers`` sets MIOPEN_DEBUG_CONV_{DIRECT,WINOGRAD,FFT,GEMM}=0 process-wide; a user
    ``file=`` script must not inherit that (it used to, but only on warm starts).  ``unrestrict`` removes exactly the
    variables the framework added and keeps a value the user exported."""
    from distributed_tensorflow_amd.utils import miopen

    for k in miopen._NON_IGEMM + ("DTF_MIOPEN_RESTRICTED", "DTF_MIOPEN_ALL_SOLVERS"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MIOPEN_DEBUG_CONV_FFT", "1")            # the user's own choice
    assert miopen.restrict_solvers()
    assert os.environ["MIOPEN_DEBUG_CONV_DIRECT"] == "0" and os.environ["MIOPEN_DEBUG_CONV_FFT"] == "1"
    miopen.unrestrict()
    assert "MIOPEN_DEBUG_CONV_DIRECT" not in os.environ and "MIOPEN_DEBUG_CONV_WINOGRAD" not in os.environ
    assert os.environ["MIOPEN_DEBUG_CONV_FFT"] == "1" and "DTF_MIOPEN_RESTRICTED" not in os.environ


def test_package_import_raises_hw_queue_count(monkeypatch):
    """A live-communicator rank uses more streams than HIP's default four hardware queues: the package raises
    GPU_MAX_HW_QUEUES to 8 before the runtime initialises, keeps a larger setting, and DTF_KEEP_HW_QUEUES=1 opts out."""
    import distributed_tensorflow_amd as pkg

    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    pkg._ensure_hw_queues()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "8"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    pkg._ensure_hw_queues()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "16"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    pkg._ensure_hw_queues()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "8"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")
    monkeypatch.setenv("DTF_KEEP_HW_QUEUES", "1")
    pkg._ensure_hw_queues()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "2"
    # a process that will replay a hipGraph puts the caller's setting back (before the HIP runtime starts)
    monkeypatch.delenv("DTF_KEEP_HW_QUEUES")
    monkeypatch.setattr(pkg, "_hw_queues_before", pkg._UNSET)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    pkg._ensure_hw_queues()
    assert os.environ["GPU_MAX_HW_QUEUES"] == "8"
    assert pkg.restore_hw_queues() is True
    assert os.environ["GPU_MAX_HW_QUEUES"] == "4"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    pkg._ensure_hw_queues()
    assert pkg.restore_hw_queues() is True and "GPU_MAX_HW_QUEUES" not in os.environ


def test_finished_jobs_are_garbage_collected_after_their_ttl(cluster):
    """The reference keeps every worker Job forever (only the ps side is reaped); a long-lived service deletes a
    finished Job ``ttlSecondsAfterFinished`` (Job spec) or ``[launcher] finished_job_ttl_seconds`` after it ended --
    with a DELETED event like any other deletion -- and keeps Jobs without a ttl, as before."""
    kept = cluster.create_namespaced_job("default", _job(0, 3, 0, "exit:0"))["metadata"]["name"]
    body = _job(1, 3, 1, "exit:0")
    body["spec"]["ttlSecondsAfterFinished"] = 0.3
    gone = cluster.create_namespaced_job("default", body)["metadata"]["name"]
    failed = _job(2, 3, 0, "exit:3")
    failed["spec"]["ttlSecondsAfterFinished"] = 0.3
    gone2 = cluster.create_namespaced_job("default", failed)["metadata"]["name"]

    def names():
        return {j["metadata"]["name"] for j in cluster.list_namespaced_job("default")["items"]}

    assert _wait(lambda: cluster.read_namespaced_job(kept, "default")["status"]["succeeded"] == 1)
    assert _wait(lambda: gone not in names() and gone2 not in names(), timeout=10)
    assert kept in names() and cluster.free_gpus() == [0, 1]
    deleted = [e["object"]["metadata"]["name"] for e in cluster.watch("default", 0, timeout=0.2) if e["type"] == "DELETED"]
    assert sorted(deleted) == sorted([gone, gone2])
    # the config-wide default applies to Jobs that carry no ttl of their own
    cluster._finished_ttl = 0.2
    late = cluster.create_namespaced_job("default", _job(3, 4, 0, "exit:0"))["metadata"]["name"]
    assert _wait(lambda: late not in names(), timeout=10) and kept in names()


def test_tasks_that_outlived_a_killed_service_are_reaped_by_the_next_one(tmp_path):
    """A service process that is SIGKILLed leaves its tasks running (each is its own session) on devices the next
    service instance would hand out again.  Every running task leaves a note (pid + kernel start time); a starting
    LocalCluster kills what the notes of a DEAD owner point to, leaves the tasks of a LIVE owner alone, and is not
    fooled by a recycled pid."""
    import signal
    import subprocess
    import sys
    import textwrap

    logs = str(tmp_path / "tasks")
    svc = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        from distributed_tensorflow_amd.utils.config import ApiConfig
        ApiConfig.override("image", "tensorflow", "tests.fake_task")
        ApiConfig.override("launcher", "verify_reclaim", "0")
        from distributed_tensorflow_amd.service import manifest
        from distributed_tensorflow_amd.service.launcher import LocalCluster
        c = LocalCluster(gpus=[0], log_dir=%r, extra_env={"PYTHONPATH": %r})
        info = {"file": "sleep:600", "resource": {"gpu": 1}, "detail": {"ps": 0, "worker": 1}}
        j = c.create_namespaced_job("default", manifest.gen_job("aaaaaaaa-0000-11f1-8000-000000000001", "worker", 0, 1, info, "", ""))
        print("task", c._jobs[("default", j["metadata"]["name"])].proc.pid, flush=True)
        time.sleep(600)
    """) % (ROOT, logs, ROOT)
    p = subprocess.Popen([sys.executable, "-c", svc], stdout=subprocess.PIPE, text=True)
    try:
        line = p.stdout.readline()
        assert line.startswith("task "), line
        task_pid = int(line.split()[1])
        ApiConfig.override("image", "tensorflow", "tests.fake_task")
        ApiConfig.override("launcher", "verify_reclaim", "0")
        ApiConfig.override("launcher", "grace_seconds", "0.3")
        # the first service is alive: a second one sharing the directory must not touch its task
        c1 = LocalCluster(gpus=[0], log_dir=logs)
        assert c1.orphans_reaped == [] and LocalCluster._proc_start(task_pid) is not None
        c1.shutdown()
        os.kill(p.pid, signal.SIGKILL)
        p.wait()
        assert LocalCluster._proc_start(task_pid) is not None          # the task outlived its service
        c2 = LocalCluster(gpus=[0], log_dir=logs)
        assert c2.orphans_reaped == ["tf-aaaaaaaa-0000-11f1-8000-000000000001-worker-0-1"]
        assert _wait(lambda: LocalCluster._proc_start(task_pid) is None, 5)
        assert os.listdir(logs + ".running") == []
        c2.shutdown()
        # a note whose pid now belongs to somebody else (start time differs): removed, nobody is killed
        with open(os.path.join(logs + ".running", "tf-x-worker-0-1.json"), "w") as f:
            json.dump({"name": "tf-x-worker-0-1", "pid": os.getpid(), "start": "1", "gpus": [0], "owner": 1, "owner_start": "0"}, f)
        c3 = LocalCluster(gpus=[0], log_dir=logs)
        assert c3.orphans_reaped == [] and os.listdir(logs + ".running") == []
        c3.shutdown()
    finally:
        if p.poll() is None:
            p.kill()
        try:
            os.killpg(task_pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError, UnboundLocalError):
            pass
        ApiConfig.override("launcher", "verify_reclaim", "1")


def test_package_front_door_lists_and_dispatches_commands(capsys):
    from distributed_tensorflow_amd import __main__ as front

    assert front.main([]) == 0
    out = capsys.readouterr().out
    assert all(w in out for w in ("serve", "cleaner | reaper", "job submit", "inspect", "evaluate", "build"))
    assert front.main(["nope"]) == 2 and "unknown command 'nope'" in capsys.readouterr().err
    assert front.main(["inspect", "/nonexistent-export"]) == 1 and "no checkpoint under" in capsys.readouterr().err


def test_small_jobs_do_not_overtake_a_job_that_waits_for_the_whole_node(cluster3):
    """No back-filling past a waiting gang: a 1-GPU job runs, a 3-GPU job waits for all three devices, and the 1-GPU jobs
    submitted after it stay Pending although two devices are free -- otherwise a steady stream of them would starve it.
    When the first job ends the big one gets the node, and the small ones follow."""
    c = cluster3
    first = c.create_namespaced_job("default", _gang("aaaaaaaa-1111-11f1-8000-000000000001", 1, file_field="sleep:1.0")[0])
    big = [c.create_namespaced_job("default", b) for b in _gang("bbbbbbbb-2222-11f1-8000-000000000002", 3, file_field="sleep:0.3")]
    small = [c.create_namespaced_job("default", _gang("cccccccc-3333-11f1-8000-00000000000%d" % i, 1, file_field="sleep:0.2")[0])
             for i in range(2)]
    assert first["status"]["phase"] == "Running" and c.free_gpus() == [1, 2]
    assert all(j["status"]["phase"] == "Pending" for j in big + small)
    names = [j["metadata"]["name"] for j in big + small]
    done = {}

    def finished():
        for n in names:
            st = c.read_namespaced_job(n, "default")["status"]
            if st["succeeded"] == 1 and n not in done:
                done[n] = st["startTime"]
        return len(done) == len(names)

    assert _wait(finished, 30), done
    assert max(done[n] for n in names[:3]) <= min(done[n] for n in names[3:])       # the big job started first
    assert _wait(lambda: c.free_gpus() == [0, 1, 2], 10)


def test_deterministic_job_option_reaches_the_kernel_dispatch():
    from distributed_tensorflow_amd.ops import conv

    kind, o = parse_file_field("builtin://resnet50?deterministic=2&steps=5")
    assert kind == "builtin" and o.deterministic == 2 and "deterministic" not in o.extra
    assert parse_file_field("builtin://resnet50?steps=5")[1].deterministic == -1
    before = (conv._DETERMINISTIC, conv._DETERMINISTIC_STRICT, os.environ.get("DTF_DETERMINISTIC"))
    try:
        assert conv.set_deterministic(2) == 2 and conv._DETERMINISTIC and conv._DETERMINISTIC_STRICT
        assert conv.set_deterministic(0) == 0 and not conv._DETERMINISTIC and not conv._DETERMINISTIC_STRICT
        with pytest.raises(ValueError):
            conv.set_deterministic(3)
    finally:
        conv._DETERMINISTIC, conv._DETERMINISTIC_STRICT = before[0], before[1]
        if before[2] is None:
            os.environ.pop("DTF_DETERMINISTIC", None)
        else:
            os.environ["DTF_DETERMINISTIC"] = before[2]
