"""Task entry point: the ``/notebooks/entry.sh`` of this framework (external X1).

The launcher runs ``python -m distributed_tensorflow_amd.service.entry`` with
exactly the argv the reference gives its container (``handlers/train_handler.py:82-83``)::

    <job_name> <task_index> <ps_hosts> <worker_hosts> <file> <data> <export> <hdfs_web> <hdfs_namenode>

and turns the TF1 cluster description into a ``torch.distributed`` rendezvous:
host names are resolved through the launcher's loopback DNS (``DTF_DNS``), rank
0's Service address becomes ``MASTER_ADDR:MASTER_PORT`` (the reference's
``tfId:2222``), ranks are ordered PS-first in the dedicated placement and are
the workers alone in the colocated one.  A colocated ``ps`` task hosts no
variables: it is the job's never-exiting coordinator process (the thing
``remove_ps.py`` exists to reap) and does not even import torch.
"""
from __future__ import annotations

import json
import os
import signal
import sys
import time


def _resolve(host, dns):
    name, _, port = host.partition(":")
    hp = dns.get(name)
    if hp:
        a, _, p = hp.partition(":")
        return a, int(p)
    return name, int(port or 2222)


def _metrics_sink(uid):
    """hset into ``<uid>:metrics`` if a store is reachable; never fatal."""
    try:
        from ..utils.store import RedisHelper

        rc = RedisHelper().getRedis()
        rc.ping()
    except Exception:
        return lambda **kw: None

    def put(**kw):
        try:
            for k, v in kw.items():
                rc.hset(uid + ":metrics", k, repr(v) if isinstance(v, float) else v)
        except Exception:
            pass

    return put


def ps_join_forever():
    """``server.join()``: block until killed."""
    signal.signal(signal.SIGTERM, lambda *a: sys.exit(143))
    while True:
        signal.pause()


def resolve_placement(requested, num_ps, env):
    """One answer per JOB: every task (ps and worker alike) derives it from the job-level facts the manifest
    exports (``DTF_JOB_WORKER_GPUS`` / ``DTF_JOB_PS_GPUS`` = ``resource.gpu`` / ``resource.ps_gpu``), never from
    its own device count -- a ps task without a GPU used to pick 'dedicated' while its workers picked 'colocated'
    and then waited 600 s for a rendezvous that could not happen.

    auto: GPU workers and no GPUs for the ps tasks -> colocated (every GPU rank is worker + PS shard host);
          ps tasks with their own GPUs, or a CPU job with ps tasks -> dedicated."""
    wg = int(env.get("DTF_JOB_WORKER_GPUS", env.get("DTF_NUM_GPUS", "0")) or 0)
    pg = int(env.get("DTF_JOB_PS_GPUS", "0") or 0)
    if requested == "dedicated" and wg > 0 and pg == 0 and num_ps >= 1:
        # GPU workers rendezvous over RCCL, a ps task without a device can only speak gloo: the group would never form
        # (a job submitted like that on the GPU box sat silent until it was killed).  Every task sees the same job-level
        # facts, so every task takes the same way out.
        print("[entry] placement=dedicated needs ps tasks with their own GPUs (resource.ps_gpu) when the workers have "
              "GPUs; this job's ps tasks have none: using the colocated placement", flush=True)
        return "colocated"
    if requested in ("colocated", "dedicated"):
        return requested
    if num_ps < 1:
        return "colocated"
    if wg > 0 and pg == 0:
        return "colocated"
    return "dedicated"


def main(argv=None):
    t_start = time.time()
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 4:
        print("usage: entry <job_name> <task_index> <ps_hosts> <worker_hosts> [file data export hdfs_web hdfs_nn]",
              file=sys.stderr)
        return 2
    argv += [""] * (9 - len(argv))
    job_name, task_index, ps_csv, worker_csv, file_field, data_dir, export_dir = argv[:7]
    task_index = int(task_index)
    ps_hosts = [h for h in ps_csv.split(",") if h]
    worker_hosts = [h for h in worker_csv.split(",") if h]
    dns = json.loads(os.environ.get("DTF_DNS", "{}"))
    task_name = os.environ.get("DTF_TASK_NAME", "%s-%d" % (job_name, task_index))
    from ..utils.naming import parse_task_name

    parsed = parse_task_name(task_name)
    uid = parsed[0] if parsed else os.environ.get("DTF_JOB_UID", "local")
    print("[entry] %s job=%s index=%d ps=%d workers=%d file=%r export=%r" % (
        task_name, job_name, task_index, len(ps_hosts), len(worker_hosts), file_field, export_dir), flush=True)

    # hdfs:// fields (the reference's entry.sh pulls `file` and `data` from HDFS through argv[8] / argv[9],
    # /root/reference/handlers/train_handler.py:80-83): fetched through the WebHDFS gateway into a task-private
    # directory before anything runs; an hdfs:// `export` is produced locally and uploaded when the task ends
    hdfs_web = argv[7]
    from ..utils import hdfs

    stage, sync = None, None
    if hdfs.is_remote(file_field) or hdfs.is_remote(data_dir) or hdfs.is_remote(export_dir):
        import tempfile

        stage = tempfile.mkdtemp(prefix="dtf_hdfs_%s_" % task_name)
        try:
            file_field = hdfs.localize(file_field, os.path.join(stage, "file"), hdfs_web)
            data_dir = hdfs.localize(data_dir, os.path.join(stage, "data"), hdfs_web)
            if hdfs.is_remote(export_dir):
                remote = hdfs.split(export_dir)[1]
                export_dir = os.path.join(stage, "export")
                os.makedirs(export_dir, exist_ok=True)
                sync = hdfs.ExportSync(hdfs.client_from(hdfs_web), export_dir, remote)
        except hdfs.HdfsError as e:
            print("[entry] %s" % e, file=sys.stderr, flush=True)
            _rmtree(stage)
            return 2
        argv[4], argv[5], argv[6] = file_field, data_dir, export_dir
        print("[entry] hdfs fields staged under %s" % stage, flush=True)
    try:
        return _main_staged(t_start, argv, job_name, task_index, ps_csv, worker_csv, file_field, data_dir, export_dir,
                            ps_hosts, worker_hosts, dns, task_name, uid, sync)
    finally:
        # the stage directory (localized file / data copies, the export mirror) lives exactly as long as the task
        _rmtree(stage)


def _rmtree(path):
    if path:
        import shutil

        shutil.rmtree(path, ignore_errors=True)


def _main_staged(t_start, argv, job_name, task_index, ps_csv, worker_csv, file_field, data_dir, export_dir,
                 ps_hosts, worker_hosts, dns, task_name, uid, sync):
    from ..utils import hdfs

    def upload_export(publish=True):
        """Mirror the export directory to hdfs:// (no-op for a local export).  -> 0 / 1."""
        if sync is None:
            return 0
        try:
            sent = sync.push(publish=publish)
            if sent:
                print("[entry] export -> hdfs:%s: %s" % (sync.remote, ", ".join(sent)), flush=True)
        except hdfs.HdfsError as e:
            print("[entry] export upload failed: %s" % e, file=sys.stderr, flush=True)
            return 1
        return 0

    def on_checkpoint(step, comm, chief):
        """Called by the trainer on EVERY rank after a checkpoint is on local disk: data first, everybody; then --
        after a barrier -- the chief publishes index + pointer."""
        if sync is None:
            return
        upload_export(publish=False)
        if comm is not None:
            comm.barrier()
        if chief:
            upload_export(publish=True)

    from ..train.jobspec import parse_file_field

    try:
        kind, spec = parse_file_field(file_field)
    except ValueError as e:
        print("[entry] %s" % e, file=sys.stderr, flush=True)
        return 2

    os.environ.update({
        "DTF_JOB_NAME": job_name, "DTF_TASK_INDEX": str(task_index), "DTF_JOB_UID": uid,
        "DTF_PS_HOSTS": ps_csv, "DTF_WORKER_HOSTS": worker_csv,
        # what Estimator-era TF scripts read instead of flags (train/session.py: ClusterSpec.from_tf_config)
        "TF_CONFIG": json.dumps({"cluster": {"ps": ps_hosts, "worker": worker_hosts},
                                 "task": {"type": job_name, "index": task_index}}),
    })

    if kind == "script":
        # user script: give it a worker-only rendezvous and run it
        if job_name == "ps":
            ps_join_forever()
        addr, port = _resolve(worker_hosts[0], dns)
        os.environ.update({"RANK": str(task_index), "WORLD_SIZE": str(len(worker_hosts)),
                           "LOCAL_RANK": "0", "MASTER_ADDR": addr, "MASTER_PORT": str(port)})
        import runpy

        from ..utils import miopen as miopen_cache

        miopen_cache.unrestrict()       # the solver restriction is for the framework's own models only
        sys.argv = [spec] + argv
        try:
            runpy.run_path(spec, run_name="__main__")
        finally:
            upload_export()         # whatever the script exported, also when it raised
        return 0

    opts = spec
    P, W = len(ps_hosts), len(worker_hosts)
    placement = resolve_placement(opts.placement, P, os.environ)

    if placement == "colocated":
        if job_name == "ps":
            print("[entry] colocated placement: ps task is the job coordinator; joining forever", flush=True)
            ps_join_forever()
        rank, world = task_index, W
        addr, port = _resolve(worker_hosts[0], dns)
    else:
        if P < 1:
            print("[entry] dedicated placement needs at least one ps task", file=sys.stderr)
            return 2
        rank = task_index if job_name == "ps" else P + task_index
        world = P + W
        addr, port = _resolve(ps_hosts[0], dns)
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0",
                       "MASTER_ADDR": addr, "MASTER_PORT": str(port)})

    put = _metrics_sink(uid)
    if getattr(opts, "graph", False) is True:
        # (graph=auto keeps the eager step's eight queues: its trial then compares the replay with the eager step as it
        # really runs, not with one whose weight-gradient stream shares a hardware queue with the compute stream)
        # a replayed whole-step hipGraph wants HIP's default hardware-queue count back (restore_hw_queues); a GPU-hot
        # warm rank initialised HIP long ago with the eager setting, where the replay measured 29 vs 21 ms: stay eager
        from .. import restore_hw_queues

        if not restore_hw_queues():
            print("[entry] graph=1 ignored on a GPU-hot rank (the HIP runtime is already up with the eager step's "
                  "hardware-queue count); running eager", flush=True)
            opts.graph = False
    from ..utils import miopen as miopen_cache

    miopen_cache.setup()
    from ..utils import gemm_tuning

    # shipped hipBLASLt / rocBLAS solution choices (read-only, no run-time tuning): +1-1.5 % per step, but 1-1.5 s of
    # solution look-up before the first step -- on only for jobs long enough to earn that back
    mode = str(opts.gemm_tuning).lower()
    if mode in ("on", "1", "true") or (mode == "auto" and opts.steps >= 10000):
        gemm_tuning.setup()
    elif mode in ("off", "0", "false") and gemm_tuning.active():
        # a GPU-hot warm worker resolved the shipped file while it was idle; this job asked for the defaults
        import torch.cuda.tunable as tunable

        tunable.enable(False)
        gemm_tuning.deactivate()
    from ..train.trainer import run_task

    if sync is not None and opts.resume:
        # resume from a remote export: the stage directory starts empty -- bring the latest checkpoint in first
        try:
            got = sync.fetch_latest()
            print("[entry] resume: %s" % ("fetched checkpoint step %s from hdfs:%s" % (got, sync.remote)
                                          if got is not None else "no checkpoint under hdfs:%s" % sync.remote), flush=True)
        except hdfs.HdfsError as e:
            print("[entry] resume: could not fetch the remote checkpoint: %s" % e, file=sys.stderr, flush=True)
            return 2
    rc = 1
    try:
        rc = run_task(job_name=job_name, task_index=task_index, num_ps=P, num_workers=W, placement=placement,
                      opts=opts, data_dir=data_dir, export_dir=export_dir, metrics=put, t_start=t_start,
                      on_checkpoint=on_checkpoint)
    except Exception as e:
        # the job record carries WHY: "worker task ... failed" alone sends the user to a task log they may not have
        put(**{"error:%s-%d" % (job_name, task_index): ("%s: %s" % (type(e).__name__, e))[:400]})
        raise
    finally:
        # also on failure: checkpoints of a run that died are the ones somebody will want.  A failed task does not
        # publish a NEW pointer (it may be ahead of peers that never wrote their shard); data files still go up.
        up = upload_export(publish=(rc == 0 and job_name == "worker" and task_index == 0))
        if rc == 0:
            rc = up
    if job_name == "ps":
        print("[entry] ps task finished serving; joining forever (waiting to be reaped)", flush=True)
        ps_join_forever()
    return rc


def leave(rc):
    """End a task process with ``rc`` as its exit status.  The task is a leaf: its exit code is the Job's
    ``status.succeeded`` / ``status.failed`` (the only thing the reference's lifecycle reads,
    ``/root/reference/eventHandlers/CleanJobHandler.py:48``), so it must reflect the training result and not the
    order in which third-party libraries (HIP runtime, RCCL, gloo) run their static destructors at interpreter
    exit.  The process group is already destroyed (``trainer.run_task``); after flushing, skip the destructors.
    ``DTF_HARD_EXIT=0`` restores a plain ``sys.exit``."""
    rc = int(rc or 0)
    if os.environ.get("DTF_HARD_EXIT", "1") == "0":
        sys.exit(rc)
    try:
        sys.stdout.flush()
        sys.stderr.flush()
        from ..utils import miopen as miopen_cache

        miopen_cache.cleanup_tagged()       # a GPU-hot warm worker's pid-keyed MIOpen copy
    finally:
        os._exit(rc)


if __name__ == "__main__":
    try:
        _rc = main()
    except SystemExit as e:         # ps_join_forever's SIGTERM handler, argparse
        _rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    except BaseException:           # noqa: BLE001 -- the traceback belongs in the task log, the code in the Job status
        import traceback

        traceback.print_exc()
        _rc = 1
    leave(_rc)
