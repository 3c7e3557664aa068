"""Manifest generator: job spec -> k8s-shaped ``v1/Service`` and ``batch/v1 Job`` dicts.

Pure functions (no I/O) so the whole contract is unit-testable
(SURVEY 4.1).  Field-for-field what ``TrainHandler.genV1Service`` /
``genV1Job`` build (reference ``handlers/train_handler.py:22-37`` and ``:60-89``):
headless Service per task selecting ``{"tf": tfId}`` on ``[k8s] headless_port``;
Job with one container named ``tfId``, ``restartPolicy: Never``, image
``[image] tensorflow`` and the nine-argument entry command.  Additions the
reference lacks (SURVEY 4 defect table): a ``resources`` stanza mapping
``resource.gpu`` onto ``amd.com/gpu`` (the reference never reads ``resource``),
``backoffLimit: 0``, and only ``ps``/``worker`` keys of ``detail`` count as task
types.
"""
from __future__ import annotations

from ..utils.config import ApiConfig
from ..utils.naming import TASK_TYPES, host_list, task_name

ENTRYPOINT = "/notebooks/entry.sh"


class SpecError(ValueError):
    """Malformed job spec -> HTTP 400 (the reference lets these become 500s)."""


def validate_spec(info):
    if not isinstance(info, dict):
        raise SpecError("job spec must be a JSON object")
    detail = info.get("detail")
    if not isinstance(detail, dict):
        raise SpecError("job spec needs a 'detail' object with 'ps' and 'worker' counts")
    counts = {}
    for k in TASK_TYPES:
        v = detail.get(k, 0)
        if isinstance(v, bool) or not isinstance(v, int) or v < 0:
            raise SpecError("detail.%s must be a non-negative integer" % k)
        counts[k] = v
    unknown = [k for k in detail if k not in TASK_TYPES]
    if unknown:
        raise SpecError("unknown task types in detail: %s" % ",".join(sorted(unknown)))
    if counts["worker"] < 1:
        raise SpecError("detail.worker must be >= 1")
    res = info.get("resource", {}) or {}
    if not isinstance(res, dict):
        raise SpecError("resource must be an object")
    gpu = res.get("gpu", 0)
    if isinstance(gpu, bool) or not isinstance(gpu, int) or gpu < 0:
        raise SpecError("resource.gpu must be a non-negative integer")
    to = info.get("timeout", 0)
    if isinstance(to, bool) or not isinstance(to, (int, float)) or to < 0:
        raise SpecError("timeout must be a non-negative number of seconds")
    return counts


def task_counts(detail):
    return [(t, int(detail.get(t, 0))) for t in TASK_TYPES]


def namespace_for(info, cfg=None):
    cfg = cfg or ApiConfig()
    return cfg.get("namespace", info.get("type", "tensorflow"), "default")


def gen_service(uid, work_type, seq, count, cfg=None):
    cfg = cfg or ApiConfig()
    tf_id = task_name(uid, work_type, seq, count)
    port = cfg.getint("k8s", "headless_port")
    return {
        "apiVersion": "v1",
        "kind": "Service",
        "metadata": {"name": tf_id},
        "spec": {
            "clusterIP": "None",
            "selector": {"tf": tf_id},
            "ports": [{"port": port, "targetPort": port}],
        },
    }


def gen_resources(resource, work_type, cfg=None):
    """``resource{cpu,mem,gpu}`` -> k8s ``resources.limits`` with ``amd.com/gpu``.
    PS tasks are host-side coordinators in the colocated placement and get no
    GPU unless ``resource.ps_gpu`` asks for one (dedicated placement)."""
    cfg = cfg or ApiConfig()
    resource = resource or {}
    limits = {}
    if "cpu" in resource:
        limits["cpu"] = str(resource["cpu"])
    if "mem" in resource:
        limits["memory"] = str(resource["mem"])
    gpu = int(resource.get("gpu", 0) or 0)
    if work_type == "ps":
        gpu = int(resource.get("ps_gpu", 0) or 0)
    if gpu:
        limits[cfg.get("launcher", "gpu_resource", "amd.com/gpu")] = gpu
    return {"limits": limits} if limits else {}


def gen_job(uid, work_type, seq, count, info, ps, workers, cfg=None):
    cfg = cfg or ApiConfig()
    tf_id = task_name(uid, work_type, seq, count)
    port = cfg.getint("k8s", "headless_port")
    container = {
        "name": tf_id,
        "image": cfg.get("image", "tensorflow"),
        "command": [
            ENTRYPOINT, work_type, str(seq), ps, workers,
            info.get("file", ""), info.get("data", "/notebooks"), info.get("export", "/tmp"),
            cfg.get("hdfs", "web"), cfg.get("hdfs", "namenode"),
        ],
        "ports": [{"containerPort": port}],
    }
    res = gen_resources(info.get("resource"), work_type, cfg)
    if res:
        container["resources"] = res
    # job-level facts every task of the job must agree on (a ps task has no device of its own to look at):
    # the placement is derived from these, never from the task's own GPU count
    r = info.get("resource") or {}
    container["env"] = [{"name": "DTF_JOB_WORKER_GPUS", "value": str(int(r.get("gpu", 0) or 0))},
                        {"name": "DTF_JOB_PS_GPUS", "value": str(int(r.get("ps_gpu", 0) or 0))}]
    spec_extra = {}
    if info.get("timeout"):
        # job-level time limit (k8s activeDeadlineSeconds); the launcher fails the task when it expires
        spec_extra["activeDeadlineSeconds"] = int(info["timeout"])
    # gang scheduling (service/launcher.py): the tasks of one job start together or not at all -- a worker that got its
    # device and waits in the rendezvous for a peer that never gets one holds the device forever
    detail = info.get("detail") or {}
    gang_gpus = int(detail.get("worker", 0) or 0) * int(r.get("gpu", 0) or 0) \
        + int(detail.get("ps", 0) or 0) * int(r.get("ps_gpu", 0) or 0)
    return {
        "apiVersion": "batch/v1",
        "kind": "Job",
        "metadata": {"name": tf_id, "annotations": {"dtf/gang": str(uid), "dtf/gang-gpus": str(gang_gpus)}},
        "spec": {
            "backoffLimit": 0,
            **spec_extra,
            "template": {
                "metadata": {"name": tf_id, "labels": {"tf": tf_id}},
                "spec": {"restartPolicy": "Never", "containers": [container]},
            },
        },
    }


def gen_all(uid, info, cfg=None):
    """-> ``(services, jobs, ps_hosts, worker_hosts)`` for a validated spec,
    PS tasks first (the response order of ``handlers/train_handler.py:126-127``)."""
    cfg = cfg or ApiConfig()
    detail = info["detail"]
    port = cfg.get("k8s", "headless_port")
    ps_hosts = host_list(uid, "ps", detail.get("ps", 0), port)
    worker_hosts = host_list(uid, "worker", detail.get("worker", 0), port)
    ps_csv, worker_csv = ",".join(ps_hosts), ",".join(worker_hosts)
    services, jobs = [], []
    for work_type, count in task_counts(detail):
        for i in range(count):
            services.append(gen_service(uid, work_type, i, count, cfg))
            jobs.append(gen_job(uid, work_type, i, count, info, ps_csv, worker_csv, cfg))
    return services, jobs, ps_hosts, worker_hosts
