"""The reference's deployment shape: three separate daemons (REST service, job watcher, reaper)
coupled only through the job store and the cluster API -- here with the native
``dtf-stored`` as the store and the node agent's k8s-shaped HTTP API + long-poll watch."""
import json
import os
import subprocess
import sys
import time
import urllib.request

import pytest

from conftest import free_port
from distributed_tensorflow_amd.native.build import spawn_store
from distributed_tensorflow_amd.utils.naming import parse_task_name
from distributed_tensorflow_amd.utils.store import StoreClient

pytestmark = pytest.mark.slow
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rest_watcher_reaper_as_separate_processes(tmp_path):
    store_proc, store_port = spawn_store()
    svc_port = free_port()
    env = dict(os.environ, PYTHONPATH=ROOT, DTF_SERVICE_PORT=str(svc_port), DTF_REDIS_PORT=str(store_port),
               DTF_REDIS_HOST="127.0.0.1", DTF_IMAGE_TENSORFLOW="tests.fake_task", DTF_LAUNCHER_GRACE_SECONDS="0.5",
               DTF_LAUNCHER_LOG_DIR=str(tmp_path / "tasks"), DTF_LOG_DIR=str(tmp_path / "logs"))
    procs = []
    try:
        for mod in ("disTf", "tf_cleaner", "remove_ps"):
            procs.append(subprocess.Popen([sys.executable, "-m", "distributed_tensorflow_amd.service." + mod],
                                          env=env, cwd=str(tmp_path), stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT))
        # wait for the REST port
        for _ in range(100):
            try:
                urllib.request.urlopen("http://127.0.0.1:%d/healthz" % svc_port, timeout=1).read()
                break
            except Exception:
                time.sleep(0.1)
        body = json.dumps({"type": "tensorflow", "file": "sleep:0.5", "detail": {"ps": 1, "worker": 2}}).encode()
        req = urllib.request.Request("http://127.0.0.1:%d/v1/train" % svc_port, data=body, method="POST")
        names = json.loads(urllib.request.urlopen(req, timeout=30).read().decode())
        uid = parse_task_name(names[0])[0]
        rc = StoreClient("127.0.0.1", store_port)
        rec = {}
        for _ in range(300):
            rec = json.loads(rc.get(uid))
            if "t_cleaned" in rec:
                break
            time.sleep(0.1)
        assert rec.get("status") == "succeeded", rec
        assert rc.smembers("success_uids") == {uid}
        assert rc.hget("tf_ps_remaining", uid) is None
        jobs = json.loads(urllib.request.urlopen(
            "http://127.0.0.1:%d/apis/batch/v1/namespaces/default/jobs" % svc_port, timeout=5).read().decode())
        assert sorted(j["metadata"]["name"] for j in jobs["items"]) == sorted(names[1:])
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            try:
                p.wait(5)
            except subprocess.TimeoutExpired:
                p.kill()
        store_proc.kill()
        store_proc.wait()
