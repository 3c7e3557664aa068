"""REST service + launcher + watcher + reaper, end to end with trivial task
processes (SURVEY 4.1 'Integration -- service' and 'Integration -- lifecycle')."""
import json
import os
import re
import time
import urllib.request

import pytest

from distributed_tensorflow_amd.service.disTf import AllInOne
from distributed_tensorflow_amd.utils.config import ApiConfig
from distributed_tensorflow_amd.utils.naming import parse_task_name


def _post(port, body, method="POST", path="/v1/train"):
    data = body if isinstance(body, bytes) else json.dumps(body).encode()
    req = urllib.request.Request("http://127.0.0.1:%d%s" % (port, path), data=data if method != "GET" else None,
                                 method=method)
    try:
        with urllib.request.urlopen(req, timeout=30) as r:
            return r.status, json.loads(r.read().decode() or "null")
    except urllib.error.HTTPError as e:
        return e.code, json.loads(e.read().decode() or "null")


def _wait(cond, timeout=20.0, every=0.05):
    t0 = time.time()
    while time.time() - t0 < timeout:
        v = cond()
        if v:
            return v
        time.sleep(every)
    return cond()


@pytest.fixture
def aio(tmp_path):
    ApiConfig.override("image", "tensorflow", "tests.fake_task")
    ApiConfig.override("launcher", "grace_seconds", "0.5")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = AllInOne(gpus=[], log_dir=str(tmp_path / "tasks"), extra_env={"PYTHONPATH": root})
    port = a.start(port=0)
    yield a, port
    a.stop()


def _alive(pid):
    try:
        os.kill(pid, 0)
        return True
    except OSError:
        return False


def test_submit_response_and_record_schema(aio):
    a, port = aio
    body = {"type": "tensorflow", "file": "sleep:30", "data": "/yyy",
            "resource": {"cpu": 2, "mem": "4g", "gpu": 0}, "detail": {"ps": 1, "worker": 2}}
    status, names = _post(port, body)
    assert status == 200
    # response: JSON array of task names without port, PS first (train_handler.py:126-127)
    assert len(names) == 3
    uid = parse_task_name(names[0])[0]
    assert names == ["tf-%s-ps-0-1" % uid, "tf-%s-worker-0-2" % uid, "tf-%s-worker-1-2" % uid]
    assert re.match(r"^[0-9a-f]{8}-[0-9a-f]{4}-1[0-9a-f]{3}-", uid)        # uuid1
    # Redis record, byte for byte (train_handler.py:129-135)
    expect = json.dumps({"ps": ["tf-%s-ps-0-1:2222" % uid],
                         "worker": ["tf-%s-worker-0-2:2222" % uid, "tf-%s-worker-1-2:2222" % uid],
                         "status": "running"})
    assert a.store.get(uid) == expect
    assert a.store.sismember("running_uids", uid)
    # every task got a Service (loopback DNS entry) and a process
    assert set(a.cluster.dns_map(uid)) == set(names)
    jobs = a.cluster.list_namespaced_job("default")["items"]
    assert sorted(j["metadata"]["name"] for j in jobs) == sorted(names)
    # watcher saw ADDED for all of them
    assert _wait(lambda: a.store.hget("tf_worker_remaining", uid) == "2")
    assert a.store.hget("tf_ps_remaining", uid) == "1"
    # GET read path
    status, rec = _post(port, None, "GET", "/v1/train?uid=" + uid)
    assert status == 200 and rec["status"] == "running" and "t_submit" in rec["metrics"]
    status, lst = _post(port, None, "GET")
    assert uid in lst["running"]
    # cancel
    status, res = _post(port, None, "DELETE", "/v1/train?uid=" + uid)
    assert res == {"cancelled": True}
    assert _wait(lambda: json.loads(a.store.get(uid)).get("t_cleaned"))
    assert a.cluster.list_namespaced_job("default")["items"] == []


def test_bad_requests_are_400_not_500(aio):
    _, port = aio
    assert _post(port, b"{not json")[0] == 400
    assert _post(port, {"type": "tensorflow"})[0] == 400
    assert _post(port, {"detail": {"ps": 1, "worker": 1, "chief": 1}})[0] == 400
    assert _post(port, None, "GET", "/v1/train?uid=nope")[0] == 404


def test_full_lifecycle_success_reaps_ps_and_services(aio):
    a, port = aio
    status, names = _post(port, {"file": "sleep:0.3", "detail": {"ps": 2, "worker": 2}})
    assert status == 200
    uid = parse_task_name(names[0])[0]
    ps_pids = [a.cluster._jobs[("default", n)].proc.pid for n in names[:2]]
    assert all(_alive(p) for p in ps_pids)
    # workers exit 0 -> delete_queue -> PsKiller deletes PS jobs+services -> DELETED -> cleaned
    rec = _wait(lambda: (lambda r: r if r.get("status") == "succeeded" else None)(json.loads(a.store.get(uid))))
    assert rec and rec["status"] == "succeeded"
    assert not any(_alive(p) for p in ps_pids), "PS tasks never exit on their own; the reaper must kill them"
    left = [j["metadata"]["name"] for j in a.cluster.list_namespaced_job("default")["items"]]
    assert sorted(left) == sorted(names[2:]), "worker Jobs stay behind as completed records (remove_ps.py:38-52)"
    assert a.cluster.dns_map(uid) == {}, "all Services deleted"
    assert a.store.hget("tf_ps_remaining", uid) is None and a.store.hget("tf_worker_remaining", uid) is None
    assert a.store.smembers("success_uids") == {uid} and not a.store.sismember("running_uids", uid)


def test_failed_worker_tears_down_whole_job(aio):
    a, port = aio
    status, names = _post(port, {"file": "failone:", "detail": {"ps": 1, "worker": 3}})
    uid = parse_task_name(names[0])[0]
    pids = [a.cluster._jobs[("default", n)].proc.pid for n in names]
    rec = _wait(lambda: (lambda r: r if r.get("t_cleaned") else None)(json.loads(a.store.get(uid))))
    assert rec and rec["status"] == "failed" and "worker" in rec["reason"] and rec["reason"].endswith("failed: exit code 3"), rec
    assert _wait(lambda: not any(_alive(p) for p in pids)), "hung survivors must be killed"
    assert a.cluster.list_namespaced_job("default")["items"] == []
    assert a.cluster.dns_map(uid) == {}
    assert a.store.smembers("failed_uids") == {uid}


def test_sigkill_escalation_for_tasks_ignoring_sigterm(aio):
    a, port = aio
    status, names = _post(port, {"file": "sleep:0.2,ignore-term", "detail": {"ps": 1, "worker": 1}})
    uid = parse_task_name(names[0])[0]
    pid = a.cluster._jobs[("default", names[0])].proc.pid
    assert _wait(lambda: json.loads(a.store.get(uid)).get("status") == "succeeded")
    assert not _alive(pid)


def test_rollback_on_partial_failure(aio):
    a, port = aio
    calls = {"n": 0}
    orig = a.cluster.create_namespaced_job

    def flaky(ns, body):
        calls["n"] += 1
        if calls["n"] == 2:
            from distributed_tensorflow_amd.service.launcher import ApiException
            raise ApiException(500, "boom")
        return orig(ns, body)

    a.cluster.create_namespaced_job = flaky
    status, res = _post(port, {"file": "sleep:5", "detail": {"ps": 1, "worker": 2}})
    assert status == 500
    assert a.cluster.list_namespaced_job("default")["items"] == []      # nothing leaked
    assert a.cluster.dns_map() == {}
    assert a.store.smembers("running_uids") == set()


def test_job_timeout_kills_hung_workers_and_cleans_up(aio):
    """`timeout` in the job spec -> activeDeadlineSeconds: hung workers are failed by the launcher,
    the failure path reaps everything."""
    a, port = aio
    status, names = _post(port, {"file": "sleep:600", "timeout": 1, "detail": {"ps": 1, "worker": 2}})
    assert status == 200
    uid = parse_task_name(names[0])[0]
    rec = _wait(lambda: (lambda r: r if r.get("t_cleaned") else None)(json.loads(a.store.get(uid))), timeout=30)
    assert rec and rec["status"] == "failed"
    assert "DeadlineExceeded" in rec["reason"], rec      # the record says WHY the task failed, not only that it did
    assert a.cluster.list_namespaced_job("default")["items"] == [] and a.cluster.dns_map(uid) == {}


def test_command_line_client_submits_waits_reports_and_cancels(aio, capsys):
    """``python -m distributed_tensorflow_amd.tools.job``: submit --wait of a job that succeeds (exit 0), of one that fails
    (exit 1, the reason printed), status listing, cancel of a running one, a refused spec (exit 1, the service's message)."""
    from distributed_tensorflow_amd.tools import job

    a, port = aio
    url = "http://127.0.0.1:%d" % port
    assert job.main(["--url", url, "submit", "--file", "sleep:0.3", "--worker", "2", "--wait", "--wait-limit", "60"]) == 0
    out = capsys.readouterr().out
    assert "uid " in out and out.count("  task tf-") == 3 and "succeeded" in out
    done_uid = out.split()[1]
    assert job.main(["--url", url, "submit", "--file", "failone:", "--worker", "3", "--wait", "--wait-limit", "60"]) == 1
    assert "failed: worker task" in capsys.readouterr().out
    assert job.main(["--url", url, "submit", "--file", "sleep:600", "--worker", "1"]) == 0
    uid = capsys.readouterr().out.split()[1]
    assert job.main(["--url", url, "status", uid]) == 0 and '"status": "running"' in capsys.readouterr().out
    assert job.main(["--url", url, "cancel", uid]) == 0
    assert job.main(["--url", url, "wait", uid, "--wait-limit", "60"]) == 1 and "cancelled" in capsys.readouterr().out
    assert job.main(["--url", url, "status"]) == 0 and '"running": []' in capsys.readouterr().out.replace("\n", "").replace("  ", " ")
    # "kubectl logs" for a task that is long gone: the node agent keeps one file per task
    assert job.main(["--url", url, "logs", done_uid, "--task", "worker-1", "--tail", "5"]) == 0
    out = capsys.readouterr().out
    assert out.count("==> tf-") == 1 and "worker-1-2 <==" in out and "fake task worker 1 sleep:0.3" in out
    assert job.main(["--url", url, "logs", done_uid]) == 0 and capsys.readouterr().out.count("==> tf-") == 3
    assert job.main(["--url", url, "logs", "no-such-uid"]) == 1
    capsys.readouterr()
    assert job.main(["--url", url, "submit", "--file", "sleep:1", "--worker", "0"]) == 1
    assert "refused: HTTP 4" in capsys.readouterr().err


def test_pod_log_through_the_k8s_shaped_api(aio):
    """``read_namespaced_pod_log`` through the HTTP client the separate daemons use: whole log, tail, 404, bad name."""
    from distributed_tensorflow_amd.service.launcher import ApiException, ClusterClient

    a, port = aio
    status, names = _post(port, {"file": "sleep:0.2", "detail": {"ps": 0, "worker": 1}})
    assert status == 200
    cc = ClusterClient("http://127.0.0.1:%d" % port)
    assert _wait(lambda: "fake task worker 0" in (cc.read_namespaced_pod_log(names[0], "default") or ""), timeout=20)
    assert cc.read_namespaced_pod_log(names[0], "default", tail_lines=1).count("\n") == 1
    assert cc.read_namespaced_pod_log(names[0], "default", tail_lines=0) == ""
    with pytest.raises(ApiException) as e:
        cc.read_namespaced_pod_log("tf-nobody-worker-0-1", "default")
    assert e.value.status == 404
    with pytest.raises(ApiException) as e:
        a.cluster.read_namespaced_pod_log("../secrets", "default")
    assert e.value.status == 400


def test_records_left_running_by_a_previous_service_instance_are_closed(tmp_path):
    """External store, service restarted: a record that says "running" while none of its tasks exists is marked failed
    with that reason when the new instance starts; a record whose tasks DO exist is left alone."""
    ApiConfig.override("image", "tensorflow", "tests.fake_task")
    ApiConfig.override("launcher", "grace_seconds", "0.5")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = AllInOne(gpus=[], log_dir=str(tmp_path / "tasks"), extra_env={"PYTHONPATH": root})
    try:
        cfg = ApiConfig()
        ghost = "99999999-0000-11f1-8000-00000000dead"
        a.store.set(ghost, json.dumps({"ps": ["tf-%s-ps-0-1:2222" % ghost], "worker": ["tf-%s-worker-0-1:2222" % ghost],
                                       "status": "running"}))
        a.store.sadd(cfg.get("redis", "running_set"), ghost)
        port = a.start(port=0)
        rec = json.loads(a.store.get(ghost))
        assert rec["status"] == "failed" and "service restarted" in rec["reason"] and "t_cleaned" in rec
        status, names = _post(port, {"file": "sleep:30", "detail": {"ps": 1, "worker": 1}})
        live = parse_task_name(names[0])[0]
        assert a.service.reconcile() == []                       # a job whose tasks exist is not touched
        assert json.loads(a.store.get(live))["status"] == "running"
        with urllib.request.urlopen("http://127.0.0.1:%d/v1/train" % port, timeout=30) as r:
            st = json.loads(r.read().decode())
        assert st["failed"] == [ghost] and st["running"] == [live]
    finally:
        a.stop()
