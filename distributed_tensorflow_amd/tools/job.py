"""Command-line client of the training service (``POST / GET / DELETE /v1/train``, the reference's one endpoint --
``/root/reference/disTf.py``, ``handlers/train_handler.py`` -- plus this framework's status and cancel).

    python -m distributed_tensorflow_amd.tools.job submit --file builtin://resnet50?steps=200 --ps 1 --worker 8 --gpu 1 \\
           --export /tmp/out [--data /data/imagenet] [--timeout 3600] [--wait]
    python -m distributed_tensorflow_amd.tools.job status [UID]
    python -m distributed_tensorflow_amd.tools.job wait UID
    python -m distributed_tensorflow_amd.tools.job cancel UID
    python -m distributed_tensorflow_amd.tools.job logs UID [--task worker-0] [--tail 50]      # "kubectl logs"

``--url`` (or ``DTF_SERVICE_URL``) names the service, default ``http://127.0.0.1:12345``.  ``wait`` exits 0 when the job
succeeded, 1 when it failed (the reason is printed), 2 when the time ran out."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import urllib.error
import urllib.request

from ..utils.naming import parse_task_name


def call(url, method, path, body=None, timeout=60):
    req = urllib.request.Request(url.rstrip("/") + path, data=None if body is None else json.dumps(body).encode(), method=method)
    try:
        with urllib.request.urlopen(req, timeout=timeout) as r:
            return r.status, json.loads(r.read().decode() or "null")
    except urllib.error.HTTPError as e:
        raw = e.read().decode()
        try:
            return e.code, json.loads(raw)
        except ValueError:
            return e.code, {"error": raw}


def wait(url, uid, limit, out=sys.stdout):
    t0, last, final = time.time(), None, None
    while limit <= 0 or time.time() - t0 < limit:
        code, rec = call(url, "GET", "/v1/train?uid=" + uid)
        if code != 200:
            print("status of %s: HTTP %d %s" % (uid, code, rec), file=sys.stderr)
            return 1
        st = rec.get("status")
        if st != last:
            print("[%6.1f s] %s%s" % (time.time() - t0, st, (": " + str(rec["reason"])) if rec.get("reason") else ""), file=out, flush=True)
            last = st
        if st in ("succeeded", "failed") and final is None:
            final = time.time()
        # "t_cleaned": the tasks are reaped and the devices back; a service whose cleaner is down still gets an answer
        if final is not None and ("t_cleaned" in rec or time.time() - final > 30):
            m = rec.get("metrics") or {}
            if m:
                print(json.dumps({k: m[k] for k in sorted(m) if not k.startswith("t_")}), file=out)
            return 0 if st == "succeeded" else 1
        time.sleep(0.5)
    print("still %s after %.0f s" % (last, limit), file=sys.stderr)
    return 2


def main(argv=None):
    ap = argparse.ArgumentParser(prog="job", description=__doc__.split("\n\n")[0])
    ap.add_argument("--url", default=os.environ.get("DTF_SERVICE_URL", "http://127.0.0.1:12345"))
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("submit")
    s.add_argument("--file", required=True, help="built-in workload (resnet50, builtin://bert_base?steps=100&batch=32) or a script path")
    s.add_argument("--ps", type=int, default=1)
    s.add_argument("--worker", type=int, default=1)
    s.add_argument("--gpu", type=int, default=None, help="devices per worker task (resource.gpu)")
    s.add_argument("--ps-gpu", type=int, default=None, help="devices per ps task (dedicated placement)")
    s.add_argument("--cpu", default=None)
    s.add_argument("--mem", default=None)
    s.add_argument("--data", default=None)
    s.add_argument("--export", default=None)
    s.add_argument("--timeout", type=int, default=None, help="job deadline in seconds (activeDeadlineSeconds)")
    s.add_argument("--wait", action="store_true")
    s.add_argument("--wait-limit", type=float, default=0, help="seconds; 0 = no limit")
    for name in ("status", "wait", "cancel", "logs"):
        p = sub.add_parser(name)
        p.add_argument("uid", nargs="?" if name == "status" else None)
        if name == "wait":
            p.add_argument("--wait-limit", type=float, default=0)
        if name == "logs":
            p.add_argument("--task", default=None, help="e.g. worker-0 or ps-0; default: every task of the job")
            p.add_argument("--tail", type=int, default=None, help="last N lines of each")
    a = ap.parse_args(argv)
    if a.cmd == "submit":
        body = {"type": "tensorflow", "file": a.file, "detail": {"ps": a.ps, "worker": a.worker}}
        res = {k: v for k, v in (("cpu", a.cpu), ("mem", a.mem), ("gpu", a.gpu), ("ps_gpu", a.ps_gpu)) if v is not None}
        if res:
            body["resource"] = res
        for k in ("data", "export", "timeout"):
            if getattr(a, k) is not None:
                body[k] = getattr(a, k)
        code, names = call(a.url, "POST", "/v1/train", body)
        if code != 200:
            print("refused: HTTP %d %s" % (code, names.get("error", names) if isinstance(names, dict) else names), file=sys.stderr)
            return 1
        uid = parse_task_name(names[0])[0]
        print("uid %s" % uid)
        for n in names:
            print("  task %s" % n)
        return wait(a.url, uid, a.wait_limit) if a.wait else 0
    if a.cmd == "status":
        code, rec = call(a.url, "GET", "/v1/train" + ("?uid=" + a.uid if a.uid else ""))
        print(json.dumps(rec, indent=1, sort_keys=True))
        return 0 if code == 200 else 1
    if a.cmd == "wait":
        return wait(a.url, a.uid, a.wait_limit)
    if a.cmd == "logs":
        q = "?uid=" + a.uid + ("&task=" + a.task if a.task else "") + ("&tail=%d" % a.tail if a.tail is not None else "")
        code, res = call(a.url, "GET", "/v1/train/logs" + q)
        if code != 200:
            print("logs of %s: HTTP %d %s" % (a.uid, code, res), file=sys.stderr)
            return 1
        for n in sorted(res):
            print("==> %s <==" % n)
            print(res[n] if res[n] is not None else "(no log)", end="" if res[n] and res[n].endswith("\n") else "\n")
        return 0
    code, rec = call(a.url, "DELETE", "/v1/train?uid=" + a.uid)
    print(json.dumps(rec))
    return 0 if code == 200 else 1


if __name__ == "__main__":
    sys.exit(main())
