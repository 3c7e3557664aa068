#!/usr/bin/env python3
"""Job submit -> first training step latency through the real control plane (north-star
metric #2), plus the teardown path: POST /v1/train to an all-in-one service, wait for the
chief's ``t_step1``, then for the lifecycle to finish (workers succeed -> watcher -> reaper kills
the PS task) and report the launcher's HBM-reclaim check.

    python scripts/submit_latency.py --model resnet50 --steps 30 --batch 256 [--fault hang:worker=0,step=5]
"""
import argparse
import json
import os
import sys
import time
import urllib.request

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.service.disTf import AllInOne  # noqa: E402
from distributed_tensorflow_amd.service.launcher import hbm_used_bytes, visible_gpu_count  # noqa: E402
from distributed_tensorflow_amd.utils.config import ApiConfig  # noqa: E402
from distributed_tensorflow_amd.utils.naming import parse_task_name  # noqa: E402


def http(port, method, path, body=None):
    req = urllib.request.Request("http://127.0.0.1:%d%s" % (port, path), method=method,
                                 data=None if body is None else json.dumps(body).encode())
    with urllib.request.urlopen(req, timeout=60) as r:
        return json.loads(r.read().decode() or "null")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--ps", type=int, default=1)
    ap.add_argument("--update-mode", default="sync")
    ap.add_argument("--placement", default="")
    ap.add_argument("--staleness", type=int, default=1)
    ap.add_argument("--warm", type=int, default=0, help="size of the launcher's warm rank pool (0 = cold starts)")
    ap.add_argument("--hot", type=int, default=0, help="1: GPU-hot warm ranks (one per free device)")
    ap.add_argument("--warm-model", default="", help='hot ranks also run this once, e.g. "resnet50?batch=256"')
    ap.add_argument("--cpu", action="store_true", help="CPU / gloo job (no amd.com/gpu request)")
    ap.add_argument("--extra", default="", help="extra query string for the job's file URI, e.g. image_size=64")
    ap.add_argument("--fault", default="")
    ap.add_argument("--cancel-after", type=float, default=0.0, help="DELETE the job this many seconds after step 1")
    ap.add_argument("--timeout", type=float, default=600)
    ap.add_argument("--out", default="")
    a = ap.parse_args()

    ApiConfig.override("launcher", "grace_seconds", "3")
    ApiConfig.override("launcher", "warm_pool", str(a.warm))
    ApiConfig.override("launcher", "warm_gpu_pool", str(a.hot))
    ApiConfig.override("launcher", "warm_model", a.warm_model)
    os.environ.setdefault("DTF_TRACE_START", "1")        # the tasks log "[start +x s] <stage>" lines
    ngpu = 0 if a.cpu else visible_gpu_count()
    gpus = list(range(min(ngpu, a.workers))) if ngpu else []
    hbm0 = hbm_used_bytes()
    aio = AllInOne(gpus=gpus, log_dir=os.path.join("gpurun_out", "tasks"))
    port = aio.start(port=0)
    if a.warm:
        t_w = time.time()
        while aio.cluster._warm is not None and aio.cluster._warm.ready() < min(a.warm, a.workers + a.ps) \
                and time.time() - t_w < 180:
            time.sleep(0.05)
        print("[submit] warm pool ready (%d) after %.1f s" % (aio.cluster._warm.ready(), time.time() - t_w), flush=True)
    if gpus:
        # the idle-device baseline of the HBM verification is probed at service start; let it finish first
        t_w = time.time()
        while not aio.cluster._idle_used and time.time() - t_w < 60:
            time.sleep(0.05)
    if a.hot and gpus:
        t_w = time.time()
        while len(aio.cluster._hot.ready()) < len(gpus) and time.time() - t_w < 240:
            time.sleep(0.05)
        print("[submit] GPU-hot ranks ready on %s after %.1f s" % (aio.cluster._hot.ready(), time.time() - t_w), flush=True)
    q = "steps=%d&batch=%d&update_mode=%s&staleness=%d&log_every=10" % (a.steps, a.batch, a.update_mode, a.staleness)
    if a.placement:
        q += "&placement=" + a.placement
    if a.extra:
        q += "&" + a.extra
    if a.fault:
        q += "&fault=" + a.fault
    spec = {"type": "tensorflow", "file": "builtin://%s?%s" % (a.model, q), "data": "/synthetic",
            "resource": {"cpu": 8, "mem": "32g", "gpu": 1 if gpus else 0}, "detail": {"ps": a.ps, "worker": a.workers}}
    t_post = time.time()
    names = http(port, "POST", "/v1/train", spec)
    t_resp = time.time()
    uid = parse_task_name(names[0])[0]
    out = {"uid": uid, "tasks": names, "post_roundtrip_s": round(t_resp - t_post, 4), "model": a.model,
           "gpus": gpus, "workers": a.workers, "ps": a.ps, "update_mode": a.update_mode, "warm_pool": a.warm,
           "gpu_hot_pool": a.hot, "warm_model": a.warm_model,
           "batch": a.batch, "force_comm": os.environ.get("DTF_FORCE_COMM", "0"),
           "idle_baseline_mb": {g: u >> 20 for g, u in aio.cluster._idle_used.items()}}
    if any(hbm0):
        out["hbm_used_before_mb"] = [b >> 20 for b in hbm0]
    print("[submit] %s -> %s" % (uid, names), flush=True)
    deadline = time.time() + a.timeout
    step1 = None
    cancelled = False
    rec = {}
    last_print = 0
    while time.time() < deadline:
        rec = http(port, "GET", "/v1/train?uid=" + uid)
        m = rec.get("metrics", {})
        if step1 is None and "t_step1" in m:
            step1 = float(m["t_step1"]) - float(m["t_submit"])
            out["submit_to_step1_s"] = round(step1, 3)
            out["submit_to_comm_ready_s"] = round(float(m["t_comm_ready"]) - float(m["t_submit"]), 3)
            out["spawn_s"] = round(float(m["t_spawned"]) - float(m["t_submit"]), 4)
            print("[submit] first step %.3f s after submit" % step1, flush=True)
            hb = hbm_used_bytes()
            if any(hb):
                out["hbm_used_during_mb"] = [b >> 20 for b in hb]
        if step1 is not None and a.cancel_after and not cancelled and time.time() - t_post > step1 + a.cancel_after:
            print("[submit] cancelling job", flush=True)
            http(port, "DELETE", "/v1/train?uid=" + uid)
            cancelled = True
        if "t_cleaned" in rec:
            break
        if time.time() - last_print > 20:
            last_print = time.time()
            print("[submit] waiting... status=%s step1=%s" % (rec.get("status"), step1), flush=True)
        time.sleep(0.1)
    out["final_status"] = rec.get("status")
    out["reason"] = rec.get("reason")
    m = rec.get("metrics", {})
    for k in ("items_per_sec", "final_loss", "step"):
        if k in m:
            out[k] = m[k]
    if "t_cleaned" in rec:
        out["submit_to_cleaned_s"] = round(float(rec["t_cleaned"]) - float(m.get("t_submit", t_post)), 3)
    # the reclaim probe of a task that ended by itself runs in its waiter thread AFTER the status event: wait for it
    n_gpu_tasks = len(gpus) if gpus else 0
    t_w = time.time()
    while len(aio.cluster.reclaim_reports) < n_gpu_tasks and time.time() - t_w < 60:
        time.sleep(0.1)
    out["hbm_reclaim"] = list(aio.cluster.reclaim_reports) or rec.get("hbm_reclaim") or []
    # start-up anatomy of the chief ("[start +x s] stage" lines, DTF_TRACE_START=1) and what a hot rank had warmed
    try:
        chief = [n for n in names if "-worker-0-" in n][0]
        with open(os.path.join("gpurun_out", "tasks", chief + ".log"), errors="replace") as f:
            lines = f.read().splitlines()
        out["chief_start_trace"] = [ln for ln in lines if ln.startswith("[start +") or ln.startswith("[warm]")
                                    or "first step done" in ln]
    except (OSError, IndexError):
        pass
    time.sleep(0.5)
    hb = hbm_used_bytes()
    if any(hb):
        out["hbm_used_after_mb"] = [b >> 20 for b in hb]
    out["jobs_left"] = [j["metadata"]["name"] for j in aio.cluster.list_namespaced_job("default")["items"]]
    out["services_left"] = list(aio.cluster.dns_map(uid))
    aio.stop()
    print(json.dumps(out), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    ok = out["final_status"] in ("succeeded",) or (bool(a.fault or a.cancel_after) and out["final_status"] == "failed")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
