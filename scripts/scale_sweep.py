#!/usr/bin/env python3
"""Scaling sweep of the headline benchmark on ONE node: bench.py under torchrun at N = 1/2/4/8 GPUs x push form
{reduce_scatter, all_to_all} x bucket size {8, 16, 32, 64} MB x whole-step hipGraph {0, 1}; one JSON table out.

    python scripts/scale_sweep.py                       # everything the node's GPU count allows
    python scripts/scale_sweep.py --gpus 1,2 --buckets 32 --graph 0,1 --steps 20
    python scripts/scale_sweep.py --device cpu --gpus 1,2 --batch 4 --image-size 32   # gloo rehearsal of the harness

Every run is exactly the driver's launch line (``python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ...``).  ``NCCL_DEBUG=INFO`` goes to a per-run log; the lines that say how many ranks the
communicator has and which algorithm / protocol / channel count RCCL picked are kept in the table, so "did RCCL really
see N ranks over xGMI" is answered by the record and not by inference.  Efficiency is value(N) / (N * value(1)) of
the same (push, bucket, graph) cell.  The script itself never touches a GPU (it only counts devices), every run is a
child process under its own time limit, and a failed or timed-out run ends the sweep (no retries on a sick node).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = re.compile(r"nranks|nRanks|Connected all|Channel \d+/\d+ :|comm 0x.* rank \d+ nranks|Algo|algorithm|"
                  r"protocol|Using network|P2P|xGMI|XGMI|NCCL version|RCCL version|Trees|Ring \d+")


def device_count():
    r = subprocess.run([sys.executable, "-c", "import torch;print(torch.cuda.device_count())"], capture_output=True,
                       text=True, timeout=300)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def run_one(n, push, bucket, graph, a, log_dir, port):
    cmd = [sys.executable]
    if n > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(a.steps), "--warmup", str(a.warmup),
            "--push", push, "--bucket-mb", str(bucket), "--graph", str(graph), "--model", a.model,
            "--device", a.device]
    if a.batch:
        cmd += ["--batch", str(a.batch)]
    if a.image_size:
        cmd += ["--image-size", str(a.image_size)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL,GRAPH")
    if n == 1 and a.force_comm:
        env["DTF_FORCE_COMM"] = "1"         # a live 1-rank group: the collectives really run
    tag = "n%d_%s_b%g_g%d" % (n, push, bucket, graph)
    log = os.path.join(log_dir, tag + ".log")
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=a.timeout)
        rc, out, err = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        rc, out, err = 124, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), \
            (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
    with open(log, "w") as f:
        f.write(err)
        f.write("\n==== stdout\n" + out)
    row = {"n_gpus": n, "push": push, "bucket_mb": bucket, "graph": graph, "rc": rc, "wall_s": round(time.time() - t0, 1),
           "log": os.path.relpath(log, ROOT)}
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    if rc == 0 and lines:
        res = json.loads(lines[-1])
        row.update(value=res["value"], unit=res["unit"], ms_per_step=res["ms_per_step"],
                   hipgraph=res["config"].get("hipgraph"), host_enqueue_ms=res["config"].get("host_enqueue_ms_per_step"),
                   final_loss=res["config"].get("final_loss"), parallelism=res["config"].get("parallelism"))
    seen, kept = set(), []
    for ln in err.splitlines():
        if "NCCL" in ln and KEEP.search(ln):
            key = re.sub(r"^\S+:\d+:\d+ \[\d+\] ", "", ln)
            key = re.sub(r"0x[0-9a-f]+", "0x..", key)
            if key not in seen and len(kept) < 24:
                seen.add(key)
                kept.append(ln.strip()[:220])
    row["rccl"] = kept
    m = re.search(r"nranks (\d+)", err) or re.search(r"nRanks (\d+)", err)
    row["rccl_nranks"] = int(m.group(1)) if m else None
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--push", default="reduce_scatter,all_to_all")
    ap.add_argument("--buckets", default="8,16,32,64")
    ap.add_argument("--graph", default="0,1")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--image-size", type=int, default=0)
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--force-comm", type=int, default=1, help="N=1: keep a live 1-rank RCCL group")
    ap.add_argument("--timeout", type=int, default=420, help="seconds per run")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scale_sweep.json"))
    a = ap.parse_args()
    have = device_count() if a.device == "cuda" else 64
    ns = [int(x) for x in a.gpus.split(",") if int(x) <= max(have, 1)]
    skipped = [int(x) for x in a.gpus.split(",") if int(x) > max(have, 1)]
    log_dir = os.path.join(os.path.dirname(a.out), "scale_sweep_logs")
    os.makedirs(log_dir, exist_ok=True)
    rows, port = [], 29700
    doc = {"what": "bench.py under torchrun: N x push x bucket MB x hipGraph", "devices_on_node": have,
           "n_skipped_not_enough_gpus": skipped, "model": a.model, "steps": a.steps, "warmup": a.warmup, "rows": rows}
    stop = False
    for n in ns:
        for push in a.push.split(","):
            for bucket in [float(b) for b in a.buckets.split(",")]:
                for graph in [int(g) for g in a.graph.split(",")]:
                    if a.device == "cpu" and graph:
                        continue
                    port += 1
                    row = run_one(n, push, bucket, graph, a, log_dir, port)
                    rows.append(row)
                    print("N=%d %-14s bucket %5g MB graph %d -> %s" % (
                        n, push, bucket, graph, ("%.1f %s, %.2f ms/step, RCCL nranks %s" % (
                            row["value"], row["unit"], row["ms_per_step"], row["rccl_nranks"]))
                        if "value" in row else "FAILED rc=%d (see %s)" % (row["rc"], row["log"])), flush=True)
                    with open(a.out, "w") as f:
                        json.dump(doc, f, indent=1)
                    if row["rc"] != 0:
                        stop = True
                    if stop:
                        break
                if stop:
                    break
            if stop:
                break
        if stop:
            break
    base = {(r["push"], r["bucket_mb"], r["graph"]): r["value"] for r in rows if r["n_gpus"] == 1 and "value" in r}
    for r in rows:
        b = base.get((r["push"], r["bucket_mb"], r["graph"]))
        if b and "value" in r:
            r["scaling_efficiency"] = round(r["value"] / (r["n_gpus"] * b), 4)
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote %s (%d runs%s)" % (a.out, len(rows), ", stopped at the first failure" if stop else ""))
    return 1 if stop else 0


if __name__ == "__main__":
    sys.exit(main())
