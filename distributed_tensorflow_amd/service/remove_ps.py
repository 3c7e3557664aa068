"""Reaper daemon (component C6, reference ``service/remove_ps.py:1-73``).

``PsKiller.run()`` blocks on ``BLPOP [event] delete_queue`` (ref ``:54-58``),
parses ``{"ps":[...], "worker":[...]}`` (ref ``:60-62``) and tears the job down:
``removePs`` deletes the Job **and** the Service of every PS task with
Foreground propagation (ref ``:19-36``) -- PS tasks never exit on their own --
and ``removeWorker`` deletes only the workers' Services, leaving the completed
worker Jobs behind as records (ref ``:38-52``).

New: a queue item with ``"kill_workers": true`` (failed job, see
``event_handlers/CleanJobHandler.py``) also deletes the worker Jobs, since the
surviving ranks of a failed job sit in an RCCL collective that will never
complete; every Job delete returns the signal that was needed and the
HBM-reclaim report of the launcher, which is appended to ``<uid>:reclaim``.
"""
from __future__ import annotations

import json
import logging
import threading

from ..utils.config import ApiConfig
from ..utils.naming import parse_task_name
from ..utils.store import RedisHelper
from .launcher import ApiException, get_cluster

log = logging.getLogger(__name__)

FOREGROUND = {"propagationPolicy": "Foreground"}


class PsKiller:
    def __init__(self, cluster=None, redis=None, cfg=None):
        self._cluster = cluster
        self._rc = redis
        self._cfg = cfg or ApiConfig()
        self._stop = threading.Event()
        self.reaped = 0

    @property
    def cluster(self):
        return self._cluster or get_cluster()

    @property
    def rc(self):
        return self._rc if self._rc is not None else RedisHelper().getRedis()

    def _record_reclaim(self, name, res):
        try:
            hbm = (res.get("details") or {}).get("hbm")
            parsed = parse_task_name(name)
            if hbm and parsed:
                self.rc.rpush("%s:reclaim" % parsed[0], json.dumps(hbm))
        except Exception:
            log.exception("could not record reclaim report for %s", name)

    def _delete_job(self, name, namespace):
        try:
            res = self.cluster.delete_namespaced_job(name, namespace, FOREGROUND)
            log.info("deleted job %s: %s", name, (res.get("details") or {}).get("signal"))
            self._record_reclaim(name, res)
        except ApiException as e:
            if e.status != 404:
                log.warning("delete job %s: %s", name, e)
        except Exception:
            log.exception("delete job %s", name)

    def _delete_service(self, name, namespace):
        try:
            self.cluster.delete_namespaced_service(name, namespace, FOREGROUND)
        except ApiException as e:
            if e.status != 404:
                log.warning("delete service %s: %s", name, e)
        except Exception:
            log.exception("delete service %s", name)

    def removePs(self, psList):
        namespace = self._cfg.get("namespace", "tensorflow")
        for ps in psList:
            self._delete_job(ps, namespace)
            self._delete_service(ps, namespace)

    def removeWorker(self, workerList, kill=False):
        namespace = self._cfg.get("namespace", "tensorflow")
        for worker in workerList:
            if kill:
                self._delete_job(worker, namespace)
            self._delete_service(worker, namespace)

    def handle(self, jsInfo):
        infoMap = json.loads(jsInfo)
        kill = bool(infoMap.get("kill_workers"))
        if kill:
            # failed job: stop the workers first so nobody keeps pushing to a
            # PS that is being torn down
            self.removeWorker(infoMap.get("worker", []), kill=True)
            self.removePs(infoMap.get("ps", []))
        else:
            self.removePs(infoMap.get("ps", []))
            self.removeWorker(infoMap.get("worker", []))
        self.reaped += 1

    def run(self, poll_timeout=1):
        rc = self.rc
        queue_key = self._cfg.get("event", "delete_queue")
        try:
            while not self._stop.is_set():
                # finite timeout so stop() is honoured; semantically BLPOP 0
                res = rc.blpop(queue_key, poll_timeout)
                if res is None:
                    continue
                try:
                    self.handle(res[1])
                except Exception:
                    log.exception("bad delete_queue item %r", res)
        except KeyboardInterrupt:
            pass

    def start(self):
        t = threading.Thread(target=self.run, name="ps-killer", daemon=True)
        t.start()
        return t

    def stop(self):
        self._stop.set()


def main():
    from ..utils.config import setup_logging

    setup_logging()
    logging.getLogger().addHandler(logging.StreamHandler())
    PsKiller().run()


if __name__ == "__main__":
    main()
