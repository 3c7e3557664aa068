"""Job submission (component C2, reference ``handlers/train_handler.py:1-146``).

:class:`TrainHandler` keeps the reference's method names and call order --
``submit`` = uid -> ``createService`` -> ``createJob`` -> ``storeInfo`` -> response
(ref ``:121-127``) -- but is independent of the web framework (tornado is not
available here; ``service/disTf.py`` mounts it on aiohttp and runs it in a
worker thread so no blocking call sits on the event loop).

Kept byte-compatible: task names, the response array (task names without port,
PS first, ref ``:126-127``), the ``<uid>`` record
``{"ps": [...], "worker": [...], "status": "running"}`` and ``SADD running_uids``
(ref ``:129-135``).  Fixed: spec validation (400 instead of a 500), one
namespace for Services and Jobs (ref ``:43`` hard-codes ``'default'``), rollback
of everything created when a later create fails (ref ``:55-58,115-118`` leak),
pipelined store writes (ref ``:133`` TODO).
"""
from __future__ import annotations

import json
import logging
import time
import uuid

from ..utils.config import ApiConfig
from ..utils.store import RedisHelper
from . import manifest
from .launcher import ApiException, get_cluster

log = logging.getLogger(__name__)


class TrainHandler:
    def __init__(self, cluster=None, redis=None, cfg=None):
        self._cluster = cluster
        self._rc = redis
        self._cfg = cfg or ApiConfig()

    @property
    def cluster(self):
        return self._cluster or get_cluster()

    @property
    def rc(self):
        return self._rc if self._rc is not None else RedisHelper().getRedis()

    # -- reference surface ---------------------------------------------------
    def parse(self, data):
        if isinstance(data, (bytes, bytearray)):
            data = data.decode()
        try:
            return json.loads(data)
        except ValueError as e:
            raise manifest.SpecError("request body is not valid JSON: %s" % e)

    def genV1Service(self, uid, workType, seq, count):
        return manifest.gen_service(uid, workType, seq, count, self._cfg)

    def genV1Job(self, uid, workType, seq, count, info, ps, workers):
        return manifest.gen_job(uid, workType, seq, count, info, ps, workers, self._cfg)

    def createService(self, uid, runInfo, namespace=None, created=None):
        namespace = namespace or self._cfg.get("namespace", "tensorflow")
        for workType, workCount in manifest.task_counts(runInfo):
            for i in range(workCount):
                body = self.genV1Service(uid, workType, i, workCount)
                try:
                    self.cluster.create_namespaced_service(namespace, body)
                    if created is not None:
                        created.append(("service", body["metadata"]["name"]))
                except ApiException as e:
                    log.info("Exception when calling CoreV1Api->create_namespaced_service: %s", e)
                    raise

    def createJob(self, uid, info, namespace=None, created=None):
        namespace = namespace or manifest.namespace_for(info, self._cfg)
        _, jobs, ps_hosts, worker_hosts = manifest.gen_all(uid, info, self._cfg)
        log.info("ps: %s", ps_hosts)
        log.info("worker: %s", worker_hosts)
        for body in jobs:
            try:
                self.cluster.create_namespaced_job(namespace, body)
                if created is not None:
                    created.append(("job", body["metadata"]["name"]))
            except ApiException as e:
                log.info("Exception when calling BatchV1Api->create_namespaced_job: %s", e)
                raise
        return ps_hosts, worker_hosts

    def storeInfo(self, uid, ps_hosts, worker_hosts, t_submit=None):
        uid = str(uid)
        info = {"ps": ps_hosts, "worker": worker_hosts, "status": "running"}
        js_info = json.dumps(info)
        rc = self.rc
        with rc.pipeline() as p:
            p.sadd(self._cfg.get("redis", "running_set"), uid)
            p.set(uid, js_info)
            if t_submit is not None:
                p.hset(uid + ":metrics", "t_submit", repr(t_submit))
                p.hset(uid + ":metrics", "t_spawned", repr(time.time()))
            p.execute()

    def _rollback(self, namespace, created):
        for kind, name in reversed(created):
            try:
                if kind == "job":
                    self.cluster.delete_namespaced_job(name, namespace)
                else:
                    self.cluster.delete_namespaced_service(name, namespace)
            except Exception as e:
                log.warning("rollback of %s %s failed: %s", kind, name, e)

    def submit(self, info, t_submit=None):
        """-> list of task names (PS first).  Raises SpecError / ApiException."""
        t_submit = t_submit or time.time()
        manifest.validate_spec(info)
        uid = uuid.uuid1()
        namespace = manifest.namespace_for(info, self._cfg)
        created = []
        try:
            # the record is written before the tasks start so that ranks can
            # report into <uid>:metrics from their first instruction
            self.rc.hset(str(uid) + ":metrics", "t_submit", repr(t_submit))
            self.createService(str(uid), info["detail"], namespace, created)
            ps_hosts, worker_hosts = self.createJob(uid, info, namespace, created)
            self.storeInfo(uid, ps_hosts, worker_hosts, t_submit)
        except Exception:
            self._rollback(namespace, created)
            self.rc.delete(str(uid) + ":metrics")
            raise
        tf_hosts = ps_hosts + worker_hosts
        return [pod.split(":")[0] for pod in tf_hosts]

    # -- read path (the reference's GET is a stub, ref :13-17) -------------------
    def status(self, uid=None):
        rc, cfg = self.rc, self._cfg
        if uid is None:
            return {
                "running": sorted(rc.smembers(cfg.get("redis", "running_set"))),
                "succeeded": sorted(rc.smembers(cfg.get("redis", "success_set"))),
                "failed": sorted(rc.smembers(cfg.get("redis", "failed_set", "failed_uids"))),
            }
        raw = rc.get(uid)
        if raw is None:
            return None
        rec = json.loads(raw)
        metrics = rc.hgetall(uid + ":metrics")
        if metrics:
            curve = metrics.pop("curve", None)
            if curve:
                try:
                    rec["curve"] = {"columns": ["step", "loss", "items_per_sec"], "points": json.loads(curve)}
                except ValueError:
                    pass
            rec["metrics"] = metrics
            try:
                rec["submit_to_step1_s"] = float(metrics["t_step1"]) - float(metrics["t_submit"])
            except (KeyError, ValueError):
                pass
        reclaim = rc.lrange(uid + ":reclaim", 0, -1)
        if reclaim:
            rec["hbm_reclaim"] = [json.loads(r) for r in reclaim]
        return rec

    def reconcile(self):
        """Called by a starting service that hosts the node agent: a record that still says "running" although none
        of its tasks exists belongs to a job of a previous service instance (killed with it, or reaped as orphans by
        ``LocalCluster.reap_orphans``) -- with an external store it would say "running" for ever.  Returns the uids
        it marked failed."""
        rc, cfg = self.rc, self._cfg
        ns = manifest.namespace_for({}, cfg)
        lost = []
        for uid in sorted(rc.smembers(cfg.get("redis", "running_set")) or []):
            uid = uid.decode() if isinstance(uid, bytes) else str(uid)
            raw = rc.get(uid)
            if raw is None:
                rc.srem(cfg.get("redis", "running_set"), uid)
                continue
            rec = json.loads(raw)
            alive = False
            for n in [h.split(":")[0] for h in rec.get("ps", []) + rec.get("worker", [])]:
                try:
                    self.cluster.read_namespaced_job(n, ns)
                    alive = True
                    break
                except ApiException:
                    pass
            if alive:
                continue
            now = time.time()
            rec.update(status="failed", reason="the service restarted while the job was running: its tasks are gone",
                       t_failed=now, t_cleaned=now)
            rc.set(uid, json.dumps(rec))
            rc.srem(cfg.get("redis", "running_set"), uid)
            rc.sadd(cfg.get("redis", "failed_set", "failed_uids"), uid)
            lost.append(uid)
        if lost:
            log.warning("jobs of a previous service instance marked failed: %s", ", ".join(lost))
        return lost

    def logs(self, uid, task=None, tail=None):
        """``{task name: its log text}`` for the job's tasks (all, or those whose name contains ``task``, e.g.
        ``worker-0``); ``None`` for an unknown uid.  A task whose log cannot be read maps to ``None``."""
        raw = self.rc.get(uid)
        if raw is None:
            return None
        rec = json.loads(raw)
        names = [h.split(":")[0] for h in rec.get("ps", []) + rec.get("worker", [])]
        out = {}
        for n in names:
            if task and ("-%s-" % task) not in n and task != n:
                continue
            try:
                out[n] = self.cluster.read_namespaced_pod_log(n, manifest.namespace_for({}, self._cfg), tail)
            except ApiException:
                out[n] = None
        return out

    def cancel(self, uid):
        """Enqueue the whole job for reaping (``DELETE /v1/train?uid=``)."""
        raw = self.rc.get(uid)
        if raw is None:
            return False
        rec = json.loads(raw)
        item = {"ps": [h.split(":")[0] for h in rec.get("ps", [])],
                "worker": [h.split(":")[0] for h in rec.get("worker", [])],
                "kill_workers": True, "reason": "cancelled"}
        self.rc.set("failed:seen:%s" % uid, "cancelled", nx=True)
        rec.update(status="failed", reason="cancelled")
        self.rc.set(uid, json.dumps(rec))
        self.rc.rpush(self._cfg.get("event", "delete_queue"), json.dumps(item))
        return True
