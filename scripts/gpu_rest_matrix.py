"""One all-in-one service on the GPU box, a sequence of REST jobs through the product path (POST /v1/train -> task ->
trainer.run_task): checkpoint + resume, stale / async update modes, hipGraph, dataset input, the other model families,
an injected worker failure.  Prints one line per job and writes gpurun_out/rest_matrix.json."""
import json, os, sys, tempfile, time, urllib.request
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributed_tensorflow_amd.service.disTf import AllInOne
from distributed_tensorflow_amd.utils.config import ApiConfig
from distributed_tensorflow_amd.utils.naming import parse_task_name


def http(port, method, path, body=None):
    req = urllib.request.Request("http://127.0.0.1:%d%s" % (port, path), data=None if body is None else json.dumps(body).encode(),
                                 method=method)
    with urllib.request.urlopen(req, timeout=30) as r:
        return json.loads(r.read().decode())


ApiConfig.override("launcher", "grace_seconds", "3")
if os.environ.get("DTF_MATRIX_HOT") == "1":        # GPU-hot warm ranks: one per free device, model hint from the last job
    ApiConfig.override("launcher", "warm_gpu_pool", "1")
    ApiConfig.override("launcher", "warm_model", "auto")
logs = os.path.join("gpurun_out", "rest_matrix_tasks")
CPU = os.environ.get("DTF_MATRIX_CPU") == "1"          # rehearsal of the script itself on a box without a device
aio = AllInOne(gpus=[] if CPU else [0], log_dir=logs)
port = aio.start(port=0)
t_w = time.time()
while not CPU and not aio.cluster._idle_used and time.time() - t_w < 60:
    time.sleep(0.05)
export = tempfile.mkdtemp(prefix="dtf-export-")
data = tempfile.mkdtemp(prefix="dtf-data-")
rng = np.random.default_rng(0)
np.save(os.path.join(data, "images.npy"), rng.integers(0, 256, (512, 224, 224, 3), dtype=np.uint8))
np.save(os.path.join(data, "labels.npy"), rng.integers(0, 1000, 512).astype(np.int64))
R = "builtin://mnist_mlp?batch=64&log_every=4&" if CPU else "builtin://resnet50?batch=64&log_every=4&"
jobs = [
    ("checkpoint every 8 steps", {"file": R + "steps=24&ckpt_every=8", "export": export}),
    ("resume to step 36", {"file": R + "steps=36&resume=1", "export": export}),
    ("stale-2", {"file": R + "steps=20&update_mode=stale&staleness=2"}),
    ("one-sided async", {"file": R + "steps=20&update_mode=async"}),
    ("hipGraph", {"file": R + "steps=20&graph=1"}),
    ("dataset input (images.npy)", {"file": R + "steps=16", "data": data}),
    ("dataset input + hipGraph", {"file": R + "steps=16&graph=1", "data": data}),
    ("inception_v3", {"file": "builtin://inception_v3?batch=32&steps=10&log_every=5"}),
    ("bert seq 128", {"file": "builtin://bert_base?batch=8&steps=10&seq_len=128&log_every=5"}),
    ("wide_deep", {"file": "builtin://wide_deep?batch=8192&steps=20&log_every=10"}),
    ("user script (examples/)", {"file": os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                                                     "train_mnist_script.py"), "export": export + "-script"}),
    ("injected failure at step 5", {"file": R + "steps=20&fault=exit:worker=0,step=5,code=17"}),
    ("after the failure", {"file": R + "steps=8"}),
]
if len(sys.argv) > 1 and sys.argv[1] == "odd":
    # option combinations nobody runs on purpose: each must end (succeed, or fail with a reason), none may hang
    M = "builtin://mnist_mlp?batch=32&log_every=5&"
    jobs = [
        ("stale-3 + hipGraph", {"file": R + "steps=20&update_mode=stale&staleness=3&graph=1"}),
        ("graph=1, 2 steps only", {"file": R + "steps=2&graph=1"}),
        ("graph=auto, 20 steps (no room for a trial)", {"file": R + "steps=20&graph=auto"}),
        ("all_to_all push, fp32 wire, 1 MB buckets", {"file": R + "steps=12&push=all_to_all&wire_dtype=fp32&bucket_mb=1"}),
        ("adam on resnet50", {"file": R + "steps=12&optimizer=adam&lr=0.001"}),
        ("resume=1 with nothing to resume", {"file": R + "steps=8&resume=1", "export": export + "-empty"}),
        ("detail ps=3 worker=1", {"file": R + "steps=8", "detail": {"ps": 3, "worker": 1}}),
        ("batch 8", {"file": "builtin://resnet50?batch=8&steps=12&log_every=4"}),
        ("wide_deep + hipGraph", {"file": "builtin://wide_deep?batch=8192&steps=20&graph=1&log_every=10"}),
        ("bert + graph=auto", {"file": "builtin://bert_base?batch=8&steps=60&seq_len=128&graph=auto&log_every=20"}),
        ("async + dedicated requested", {"file": M + "steps=20&update_mode=async&placement=dedicated"}),
        ("dataset smaller than a batch", {"file": "builtin://resnet50?batch=1024&steps=4", "data": data}),
        ("unknown model option", {"file": R + "steps=6&no_such_option=1"}),
        ("inception_v3 aux_logits", {"file": "builtin://inception_v3?batch=16&steps=6&aux_logits=1&log_every=3"}),
        ("dataset + graph=auto, batch 32", {"file": "builtin://resnet50?batch=32&steps=80&graph=auto&log_every=20", "data": data}),
        ("hipGraph + checkpoint every 20", {"file": "builtin://resnet50?batch=32&steps=60&graph=1&ckpt_every=20&log_every=20",
                                            "export": export + "-g"}),
        ("resume of it, graph=auto", {"file": "builtin://resnet50?batch=32&steps=120&graph=auto&resume=1&log_every=20",
                                      "export": export + "-g"}),
        ("same 120 steps without graph or resume", {"file": "builtin://resnet50?batch=32&steps=120&log_every=20"}),
        ("wide_deep + graph=auto", {"file": "builtin://wide_deep?batch=8192&steps=80&graph=auto&log_every=20"}),
        ("inception_v3 + hipGraph", {"file": "builtin://inception_v3?batch=16&steps=20&graph=1&log_every=10"}),
    ]
    if len(sys.argv) > 2:
        jobs = jobs[-int(sys.argv[2]):]
if len(sys.argv) > 1 and sys.argv[1] == "cost":
    # what does the default (graph=auto for a one-worker GPU job) cost or gain a job, start to end?  items/s of the whole loop
    B = "builtin://resnet50?log_every=50&"
    jobs = [(("batch %d, %d steps, graph=%s" % (b, n, g)), {"file": B + "batch=%d&steps=%d&graph=%s" % (b, n, g)})
            for b, n in ((256, 60), (256, 300), (64, 300), (32, 300)) for g in ("0", "default")]
if len(sys.argv) > 1 and sys.argv[1] == "strict":
    # the job option deterministic=2 at a batch size whose default-mode step differs run to run
    B3 = "builtin://resnet50?batch=32&steps=60&log_every=20&graph=0&"
    jobs = [("batch 32, default mode, run %d" % i, {"file": B3[:-1]}) for i in (1, 2)] + \
           [("batch 32, deterministic=2, run %d" % i, {"file": B3 + "deterministic=2"}) for i in (1, 2)] + \
           [("batch 32, deterministic=2, hipGraph", {"file": B3.replace("graph=0", "graph=1") + "deterministic=2"})]
if len(sys.argv) > 1 and sys.argv[1] == "lr":
    # learning-rate schedule (warm-up + cosine) eagerly, replayed, and interrupted + resumed: one final loss
    L = "builtin://resnet50?batch=64&log_every=20&lr_warmup_steps=10&lr_schedule=cosine&lr_final_fraction=0.05&"
    jobs = [
        ("constant lr, 60 steps", {"file": "builtin://resnet50?batch=64&log_every=20&steps=60&graph=0"}),
        ("warm-up + cosine, eager", {"file": L + "steps=60&graph=0"}),
        ("warm-up + cosine, hipGraph", {"file": L + "steps=60&graph=1"}),
        ("warm-up + cosine, 30 steps of 60 + checkpoint", {"file": L + "steps=60&graph=0&fault=exit:worker=0,step=31,code=9&ckpt_every=30",
                                                           "export": export + "-lr"}),
        ("resumed to 60 (default graph=auto)", {"file": L + "steps=60&resume=1", "export": export + "-lr"}),
    ]
if len(sys.argv) > 1 and sys.argv[1] == "resume":
    # interrupted + resumed == uninterrupted, eagerly and replayed (batch 64: the smallest size whose step is reproducible
    # run to run in the default mode); and the graph trial against the eager loop for Wide&Deep
    R6 = "builtin://resnet50?batch=64&log_every=20&"
    W8 = "builtin://wide_deep?batch=8192&log_every=20&"
    jobs = [
        ("60 steps, eager", {"file": R6 + "steps=60"}),
        ("30 steps + checkpoint, eager", {"file": R6 + "steps=30", "export": export + "-e"}),
        ("resumed to 60, eager", {"file": R6 + "steps=60&resume=1", "export": export + "-e"}),
        ("60 steps, hipGraph", {"file": R6 + "steps=60&graph=1"}),
        ("30 steps + checkpoint, hipGraph", {"file": R6 + "steps=30&graph=1", "export": export + "-h"}),
        ("resumed to 60, hipGraph", {"file": R6 + "steps=60&resume=1&graph=1", "export": export + "-h"}),
        ("wide_deep 80 steps, eager", {"file": W8 + "steps=80"}),
        ("wide_deep 80 steps, hipGraph", {"file": W8 + "steps=80&graph=1"}),
        ("wide_deep 80 steps, graph=auto", {"file": W8 + "steps=80&graph=auto"}),
    ]
if len(sys.argv) > 1 and sys.argv[1] == "edges":
    # control-plane edges on the real device: two jobs at once on one GPU, a cancel mid-run, a deadline, an unsatisfiable gang
    import urllib.error

    def post(spec):
        body = {"type": "tensorflow", "data": "/synthetic", "resource": {"cpu": 8, "mem": "32g", "gpu": 1}, "detail": {"ps": 1, "worker": 1}}
        body.update(spec)
        return parse_task_name(http(port, "POST", "/v1/train", body)[0])[0]

    def wait(uid, limit=150):
        t0 = time.time()
        while time.time() - t0 < limit:
            rec = http(port, "GET", "/v1/train?uid=" + uid)
            if "t_cleaned" in rec:
                return rec
            time.sleep(0.2)
        return dict(rec, status="HUNG")

    ok = True
    t0 = time.time()
    a = post({"file": R + "steps=300"})
    b = post({"file": R + "steps=40"})
    time.sleep(1.0)
    phases = {("A" if a in j["metadata"]["name"] else "B") + "-" + j["metadata"]["name"].split("-")[-3]: j["status"]["phase"]
              for j in aio.cluster.list_namespaced_job(ApiConfig().get("namespace", "tensorflow"))["items"]}
    ra, rb = wait(a), wait(b)
    ta, tb = float(ra["metrics"]["t_step1"]), float(rb["metrics"]["t_step1"])
    print("two jobs at once on one GPU: phases after 1 s %s; A %s, B %s; B's first step %.1f s after A's (A ran %s steps first)" % (
        phases, ra["status"], rb["status"], tb - ta, ra["metrics"].get("step")), flush=True)
    ok &= ra["status"] == "succeeded" and rb["status"] == "succeeded" and tb > ta
    c = post({"file": R + "steps=100000&log_every=50"})
    while "t_step1" not in http(port, "GET", "/v1/train?uid=" + c).get("metrics", {}):
        time.sleep(0.1)
    time.sleep(2.0)
    t1 = time.time()
    http(port, "DELETE", "/v1/train?uid=" + c)
    rc_ = wait(c)
    print("cancel mid-run: %s (%s) %.1f s after DELETE, step %s reached, HBM reclaimed %s" % (
        rc_["status"], rc_.get("reason"), time.time() - t1, rc_["metrics"].get("step"),
        [r.get("reclaimed") for r in rc_.get("hbm_reclaim") or []]), flush=True)
    ok &= rc_["status"] == "failed" and rc_.get("reason") == "cancelled"
    d = post({"file": R + "steps=50&fault=hang:worker=0,step=5", "timeout": 12})
    rd = wait(d)
    print("hung worker with timeout=12: %s (%s) after %.1f s, HBM reclaimed %s" % (
        rd["status"], rd.get("reason"), float(rd.get("t_failed", 0)) - float(rd["metrics"]["t_submit"]),
        [r.get("reclaimed") for r in rd.get("hbm_reclaim") or []]), flush=True)
    ok &= rd["status"] == "failed"
    try:
        post({"file": R + "steps=5", "detail": {"ps": 1, "worker": 2}})
        print("two 1-GPU workers on a 1-GPU node: ACCEPTED (wrong)")
        ok = False
    except urllib.error.HTTPError as e:
        print("two 1-GPU workers on a 1-GPU node: HTTP %d %s" % (e.code, e.read().decode()), flush=True)
    e_ = post({"file": R + "steps=8"})
    re_ = wait(e_)
    t_f = time.time()
    while aio.cluster.free_gpus() != ([] if CPU else [0]) and time.time() - t_f < 20:
        time.sleep(0.1)         # the finished worker's device comes back once the HBM probe has seen it empty
    print("a plain job afterwards: %s, step %s, free devices %s, reservations %s" % (
        re_["status"], re_["metrics"].get("step"), aio.cluster.free_gpus(), aio.cluster._gangs), flush=True)
    ok &= re_["status"] == "succeeded" and aio.cluster.free_gpus() == ([] if CPU else [0])
    aio.stop()
    print("edges:", "all as intended" if ok else "SOMETHING WRONG", "(%.0f s)" % (time.time() - t0))
    sys.exit(0 if ok else 1)
if len(sys.argv) > 1 and sys.argv[1] == "burst":
    # 24 jobs POSTed from 8 threads at once onto one device: each is a gang of (ps, worker), all must run one after the
    # other, none may be lost, the device and every reservation must be back at the end
    import threading

    N = 24
    uids, errs, lock = [], [], threading.Lock()

    def submit(i):
        body = {"type": "tensorflow", "data": "/synthetic", "resource": {"cpu": 8, "mem": "32g", "gpu": 1},
                "detail": {"ps": 1, "worker": 1},
                "file": (R + "steps=30") if i % 3 else "builtin://mnist_mlp?batch=64&steps=200&log_every=100"}
        try:
            u = parse_task_name(http(port, "POST", "/v1/train", body)[0])[0]
            with lock:
                uids.append(u)
        except Exception as e:      # noqa: BLE001
            with lock:
                errs.append(repr(e))

    t0 = time.time()
    ths = [threading.Thread(target=submit, args=(i,)) for i in range(N)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print("%d jobs accepted in %.2f s (%d refused: %s)" % (len(uids), time.time() - t0, len(errs), errs[:2]), flush=True)
    peak_running = 0
    t_say = time.time()
    while time.time() - t0 < 600:
        st = http(port, "GET", "/v1/train")
        ns = ApiConfig().get("namespace", "tensorflow")
        running = sum(1 for j in aio.cluster.list_namespaced_job(ns)["items"]
                      if j["status"]["phase"] == "Running" and "-worker-" in j["metadata"]["name"])
        peak_running = max(peak_running, running)
        if not st["running"]:
            break
        if time.time() - t_say > 20:
            t_say = time.time()
            print("   ... %d running, %d succeeded, %d failed after %.0f s" % (
                len(st["running"]), len(st["succeeded"]), len(st["failed"]), time.time() - t0), flush=True)
        time.sleep(0.25)
    st = http(port, "GET", "/v1/train")
    recs = [http(port, "GET", "/v1/train?uid=" + u) for u in uids]
    t_f = time.time()
    while aio.cluster.free_gpus() != ([] if CPU else [0]) and time.time() - t_f < 30:
        time.sleep(0.1)
    firsts = sorted(float(r["metrics"]["t_step1"]) for r in recs if "t_step1" in r.get("metrics", {}))
    gaps = [b - a for a, b in zip(firsts, firsts[1:])]
    ok = (len(uids) == N and len(st["succeeded"]) == N and not st["failed"] and not st["running"]
          and aio.cluster.free_gpus() == ([] if CPU else [0]) and not aio.cluster._gangs and (CPU or peak_running == 1))
    print("burst of %d: %d succeeded, %d failed, %d still running after %.0f s; worker tasks running at once: at most %d; "
          "first steps %.1f s apart on average; free devices %s, reservations %s, torn-down gangs %s" % (
              N, len(st["succeeded"]), len(st["failed"]), len(st["running"]), time.time() - t0, peak_running,
              sum(gaps) / max(1, len(gaps)), aio.cluster.free_gpus(), aio.cluster._gangs, len(aio.cluster._gang_broken)), flush=True)
    rcl = [x.get("reclaimed") for r in recs for x in (r.get("hbm_reclaim") or [])]
    print("HBM reclaim reports: %d, all reclaimed: %s" % (len(rcl), all(rcl)), flush=True)
    aio.stop()
    print("burst:", "all as intended" if ok else "SOMETHING WRONG")
    sys.exit(0 if ok else 1)
out = []
for name, spec in jobs:
    body = {"type": "tensorflow", "data": "/synthetic", "resource": {"cpu": 8, "mem": "32g", "gpu": 1}, "detail": {"ps": 1, "worker": 1}}
    body.update(spec)
    t0 = time.time()
    names = http(port, "POST", "/v1/train", body)
    uid = parse_task_name(names[0])[0]
    rec = {}
    t_say = time.time()
    while time.time() - t0 < 150:
        rec = http(port, "GET", "/v1/train?uid=" + uid)
        if "t_cleaned" in rec:
            break
        if time.time() - t_say > 30:
            t_say = time.time()
            print("   ... %s still %s after %.0f s" % (name, rec.get("status"), time.time() - t0), flush=True)
        time.sleep(0.2)
    else:
        print("   !!! %s did not finish in 150 s: cancelling" % name, flush=True)
        http(port, "DELETE", "/v1/train?uid=" + uid)
        time.sleep(8)
        rec = http(port, "GET", "/v1/train?uid=" + uid)
        rec["status"] = "HUNG (cancelled)"
    m = rec.get("metrics", {})
    log = ""
    for f in sorted(os.listdir(logs)):
        if uid in f and "worker" in f:
            log = open(os.path.join(logs, f), errors="replace").read()
    notes = [k for k in ("resumed from step", "captured in a hipGraph", "input:", "wrote checkpoint", "injecting", "steps on cuda",
                         "-> replaying", "-> staying eager", "ignored") if k in log]
    rcl = rec.get("hbm_reclaim") or []
    row = {"job": name, "status": rec.get("status"), "reason": rec.get("reason"), "step": m.get("step"),
           "first_loss": m.get("first_loss"), "final_loss": m.get("final_loss"), "items_per_sec": m.get("items_per_sec"),
           "seconds": round(time.time() - t0, 1), "log_notes": notes,
           "hbm_reclaimed": [r.get("reclaimed") for r in rcl]}
    out.append(row)
    def g4(v):
        try:
            return ("%.10g" if len(sys.argv) > 1 and sys.argv[1] in ("strict", "lr") else "%.4g") % float(v)        # (cutting the string cut the exponent off: 7.04e-06 read as 7.04)
        except (TypeError, ValueError):
            return str(v)

    if len(sys.argv) > 1 and sys.argv[1] == "cost":
        print("%-38s %8.0f items/s over the whole loop" % (name, float(row["items_per_sec"] or 0)), flush=True)
    print("%-30s %-10s step %-4s loss %s -> %s  %s  reclaimed %s  %.1f s %s" % (
        name, row["status"], row["step"], g4(row["first_loss"]), g4(row["final_loss"]), notes, row["hbm_reclaimed"],
        row["seconds"], ("(" + str(row["reason"]) + ")") if row["reason"] else ""), flush=True)
aio.stop()
odd = len(sys.argv) > 1 and sys.argv[1] in ("odd", "resume", "cost", "strict", "lr")
json.dump(out, open(os.path.join("gpurun_out", "rest_matrix_odd.json" if odd else "rest_matrix.json"), "w"), indent=1)
if odd:
    bad = [r["job"] for r in out if "HUNG" in str(r["status"]) or (r["status"] != "succeeded" and not r["reason"])]
else:
    bad = [r["job"] for r in out if (r["status"] != "succeeded") != ("failure" in r["job"] and "after" not in r["job"])]
print("unexpected outcomes:", bad)
sys.exit(1 if bad else 0)
