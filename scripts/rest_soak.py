"""A long REST job on the GPU: ResNet-50 batch 64 from a dataset directory, a checkpoint every 1 000 steps, N steps (default
12 000).  Samples the worker process (host RSS, open files, threads) and the device memory while it runs; all must be flat,
the throughput of the last third must match the first third's, every checkpoint must be complete.  gpurun_out/rest_soak.json"""
import json, os, sys, tempfile, time, urllib.request
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
from distributed_tensorflow_amd.parallel import checkpoint as ckpt
from distributed_tensorflow_amd.service.disTf import AllInOne
from distributed_tensorflow_amd.utils.config import ApiConfig
from distributed_tensorflow_amd.utils.naming import parse_task_name

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12000


def http(port, method, path, body=None):
    req = urllib.request.Request("http://127.0.0.1:%d%s" % (port, path), data=None if body is None else json.dumps(body).encode(), method=method)
    with urllib.request.urlopen(req, timeout=30) as r:
        return json.loads(r.read().decode())


ApiConfig.override("launcher", "grace_seconds", "3")
logs = os.path.join("gpurun_out", "rest_soak_tasks")
aio = AllInOne(gpus=[0], log_dir=logs)
port = aio.start(port=0)
t_w = time.time()
while not aio.cluster._idle_used and time.time() - t_w < 60:
    time.sleep(0.05)
export = tempfile.mkdtemp(prefix="dtf-soak-export-")
data = tempfile.mkdtemp(prefix="dtf-soak-data-")
rng = np.random.default_rng(0)
np.save(os.path.join(data, "images.npy"), rng.integers(0, 256, (1024, 224, 224, 3), dtype=np.uint8))
np.save(os.path.join(data, "labels.npy"), rng.integers(0, 1000, 1024).astype(np.int64))
body = {"type": "tensorflow", "file": "builtin://resnet50?batch=64&steps=%d&ckpt_every=1000&log_every=500&lr=0.01" % steps, "data": data,
        "export": export, "resource": {"cpu": 8, "mem": "32g", "gpu": 1}, "detail": {"ps": 1, "worker": 1}}
t0 = time.time()
names = http(port, "POST", "/v1/train", body)
uid = parse_task_name(names[0])[0]
samples, rec, worker = [], {}, None
while time.time() - t0 < 1000:
    rec = http(port, "GET", "/v1/train?uid=" + uid)
    if "t_cleaned" in rec:
        break
    if worker is None:
        for p in psutil.process_iter(["cmdline"]):
            c = " ".join(p.info["cmdline"] or [])
            if uid in c and "worker" in c:
                worker = p
    elif worker.is_running():
        try:
            with worker.oneshot():
                samples.append({"t": round(time.time() - t0, 1), "rss_mb": worker.memory_info().rss >> 20, "threads": worker.num_threads(),
                                "fds": worker.num_fds(), "step": rec.get("metrics", {}).get("step")})
        except psutil.Error:
            pass
    time.sleep(5)
log = ""
for f in sorted(os.listdir(logs)):
    if uid in f and "worker" in f:
        log = open(os.path.join(logs, f), errors="replace").read()
import re
rates = [float(x) for x in re.findall(r"step \d+ loss [0-9.naninf]+\s+([0-9.]+) items/s", log)]
latest = ckpt.latest_step(export)
files = sorted(os.listdir(export))
out = {"status": rec.get("status"), "metrics": rec.get("metrics"), "seconds": round(time.time() - t0, 1), "samples": samples,
       "cumulative_items_per_s_in_the_log": rates, "latest_checkpoint": latest, "export_files": len(files),
       "export_mb": sum(os.path.getsize(os.path.join(export, f)) for f in files) >> 20}
json.dump(out, open(os.path.join("gpurun_out", "rest_soak.json"), "w"), indent=1)
aio.stop()
s = [x for x in samples if x["t"] > 30]
print("status %s after %.0f s; latest checkpoint %s; %d files, %d MB in export" % (out["status"], out["seconds"], latest, len(files), out["export_mb"]))
if s:
    print("worker RSS %d -> %d MB, threads %d -> %d, fds %d -> %d over %d samples" % (s[0]["rss_mb"], s[-1]["rss_mb"], s[0]["threads"],
                                                                                 s[-1]["threads"], s[0]["fds"], s[-1]["fds"], len(s)))
print("items/s (cumulative, from the log):", rates[:3], "...", rates[-3:])
ok = out["status"] == "succeeded" and latest == steps and (not s or (s[-1]["rss_mb"] - s[0]["rss_mb"] < 300 and s[-1]["fds"] - s[0]["fds"] < 16))
sys.exit(0 if ok else 1)
