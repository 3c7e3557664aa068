"""Manifest generator = pure function of the job spec (SURVEY 4.1 'Unit -- control plane')."""
import json
import os

import pytest

from distributed_tensorflow_amd.service import manifest
from distributed_tensorflow_amd.utils.naming import parse_task_name, search_pattern, task_name

UID = "3e8f3702-d10b-11e8-abe4-fa163ef8da8a"
POST_DATA = os.path.join(os.path.dirname(__file__), "..", "post-data")


def sample():
    """The reference's canonical body (``post-data:1-14``), field for field."""
    return {"type": "tensorflow", "file": "/xxx/", "data": "/yyy",
            "resource": {"cpu": 2, "mem": "4g", "gpu": 1}, "detail": {"ps": 1, "worker": 2}}


def test_shipped_sample_body_is_valid():
    with open(POST_DATA) as f:
        info = json.load(f)
    assert manifest.validate_spec(info) == {"ps": 1, "worker": 8}


def test_task_naming_round_trip():
    n = task_name(UID, "ps", 0, 1)
    assert n == "tf-3e8f3702-d10b-11e8-abe4-fa163ef8da8a-ps-0-1"   # CleanJobHandler.py:20
    assert parse_task_name(n) == (UID, "ps", 0, 1)
    assert parse_task_name(task_name(UID, "worker", 11, 16)) == (UID, "worker", 11, 16)
    assert search_pattern("worker", n) == (None, None, None)
    assert search_pattern("ps", n) == (UID, "0", "1")
    assert parse_task_name("tf-nope") is None


def test_service_manifest_matches_reference_fields():
    s = manifest.gen_service(UID, "worker", 1, 2)
    assert s["apiVersion"] == "v1" and s["kind"] == "Service"
    assert s["metadata"]["name"] == "tf-%s-worker-1-2" % UID
    assert s["spec"]["clusterIP"] == "None"                          # headless
    assert s["spec"]["selector"] == {"tf": s["metadata"]["name"]}
    assert s["spec"]["ports"] == [{"port": 2222, "targetPort": 2222}]


def test_job_manifest_argv_and_gpu_resources():
    info = sample()
    ps = "tf-%s-ps-0-1:2222" % UID
    workers = "tf-%s-worker-0-2:2222,tf-%s-worker-1-2:2222" % (UID, UID)
    j = manifest.gen_job(UID, "worker", 1, 2, info, ps, workers)
    assert j["apiVersion"] == "batch/v1" and j["kind"] == "Job"
    tpl = j["spec"]["template"]
    assert tpl["metadata"]["labels"] == {"tf": j["metadata"]["name"]}
    assert tpl["spec"]["restartPolicy"] == "Never"
    c = tpl["spec"]["containers"][0]
    assert c["name"] == j["metadata"]["name"]
    # argv order of handlers/train_handler.py:82-83
    assert c["command"][:8] == ["/notebooks/entry.sh", "worker", "1", ps, workers, "/xxx/", "/yyy", "/tmp"]
    assert len(c["command"]) == 10
    assert c["ports"] == [{"containerPort": 2222}]
    # resource.gpu -> amd.com/gpu (the reference ignores `resource` entirely)
    assert c["resources"]["limits"]["amd.com/gpu"] == 1
    assert c["resources"]["limits"]["cpu"] == "2" and c["resources"]["limits"]["memory"] == "4g"
    # PS tasks are host-side coordinators: no GPU
    jp = manifest.gen_job(UID, "ps", 0, 1, info, ps, workers)
    assert "amd.com/gpu" not in jp["spec"]["template"]["spec"]["containers"][0].get("resources", {}).get("limits", {})


def test_gen_all_orders_ps_first():
    services, jobs, ps_hosts, worker_hosts = manifest.gen_all(UID, sample())
    assert [s["metadata"]["name"] for s in services] == [j["metadata"]["name"] for j in jobs]
    assert ps_hosts == ["tf-%s-ps-0-1:2222" % UID]
    assert worker_hosts == ["tf-%s-worker-0-2:2222" % UID, "tf-%s-worker-1-2:2222" % UID]
    assert [j["metadata"]["name"] for j in jobs] == [h.split(":")[0] for h in ps_hosts + worker_hosts]


@pytest.mark.parametrize("bad", [
    [], {"type": "tensorflow"}, {"detail": {"ps": -1, "worker": 1}}, {"detail": {"ps": 1, "worker": 0}},
    {"detail": {"ps": 1, "worker": 1, "evaluator": 1}},     # any key became a task type in the reference
    {"detail": {"ps": "1", "worker": 1}}, {"detail": {"worker": 1}, "resource": {"gpu": -2}},
])
def test_validation_rejects_malformed_specs(bad):
    with pytest.raises(manifest.SpecError):
        manifest.validate_spec(bad)


def test_validation_accepts_sample():
    assert manifest.validate_spec(sample()) == {"ps": 1, "worker": 2}
