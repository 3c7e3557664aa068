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
    c = LocalCluster(gpus=[0, 1], log_dir=str(tmp_path), extra_env={"PYTHONPATH": ROOT})
    c._verify_reclaim = lambda t: None        # no real devices here
    yield c
    c.shutdown()


def _job(seq, count, gpu, file_field="sleep:0.4"):
    info = {"file": file_field, "resource": {"gpu": gpu}, "detail": {"ps": 0, "worker": count}}
    return manifest.gen_job(UID, "worker", seq, count, info, "", "")


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
