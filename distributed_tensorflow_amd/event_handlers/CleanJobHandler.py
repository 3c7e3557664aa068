"""Job-completion state machine (component C5, reference
``eventHandlers/CleanJobHandler.py:1-80``).

Same Redis keys and the same happy path as the reference:

* ADDED    -> ``HSETNX [event] ps_key|worker_key uid count``            (ref ``:22-31``)
* MODIFIED -> worker ``succeeded == 1``: ``HINCRBY worker_key uid -1``; at 0
  ``RPUSH [event] delete_queue {"ps":[...], "worker":[...]}``           (ref ``:37-67``)
* DELETED  -> PS job: ``HINCRBY ps_key uid -1``; at 0 ``HDEL`` both hashes (ref ``:69-80``)

and the defects listed in SURVEY 4 are closed:

* every decrement is guarded by ``SADD <key>:seen:<uid> <seq>`` so replayed or
  repeated events (watch restart, several MODIFIED with ``succeeded == 1``)
  never double-decrement;
* MODIFIED/DELETED run the ADDED ``HSETNX`` first (the count is in the task
  name), so out-of-order delivery cannot create a counter at -1;
* a failed worker marks the job ``failed`` and enqueues the *whole* job for
  reaping (``"kill_workers": true``) -- the reference's ``# TODO failed``
  (``:65-67``) leaks PS pods, Services and keys forever;
* at the end the ``<uid>`` record's ``status`` becomes ``succeeded``/``failed``
  and the uid moves ``running_uids -> success_uids|failed_uids`` -- the
  reference's ``# TODO record successful tfId`` (``:77``).
"""
import json
import logging
import time

from ..utils.config import ApiConfig
from ..utils.naming import search_pattern, task_name
from ..utils.store import RedisHelper
from .EventHandler import EventHandler

log = logging.getLogger(__name__)


def _get(status, key):
    if status is None:
        return None
    if isinstance(status, dict):
        return status.get(key)
    return getattr(status, key, None)


class CleanJobHandler(EventHandler):
    def __init__(self, redis=None, cfg=None):
        self._rc = redis
        self._cfg = cfg

    # -- plumbing -----------------------------------------------------------
    @property
    def rc(self):
        return self._rc if self._rc is not None else RedisHelper().getRedis()

    @property
    def cfg(self):
        return self._cfg if self._cfg is not None else ApiConfig()

    def _key(self, name):
        return self.cfg.get("event", name)

    def searchPattern(self, workType, name):
        return search_pattern(workType, name)

    def _register(self, rc, workType, tfId, cnt):
        rc.hsetnx(self._key("ps_key" if workType == "ps" else "worker_key"), tfId, cnt)

    def _first_time(self, rc, what, tfId, seq):
        return rc.sadd("%s:seen:%s" % (what, tfId), seq) == 1

    def _update_record(self, rc, tfId, **fields):
        raw = rc.get(tfId)
        try:
            rec = json.loads(raw) if raw else {}
        except ValueError:
            rec = {}
        rec.update(fields)
        rc.set(tfId, json.dumps(rec))
        return rec

    def _is_final(self, rc, tfId):
        """Late events for a job that is already cleaned up (a worker that dies
        from the reaper's SIGTERM, a replayed watch) must not resurrect it."""
        raw = rc.get(tfId)
        try:
            return bool(raw) and "t_cleaned" in json.loads(raw)
        except ValueError:
            return False

    # -- ADDED --------------------------------------------------------------
    def addEvent(self, objName, eStatus):
        rc = self.rc
        for workType in ("ps", "worker"):
            tfId, _, cnt = self.searchPattern(workType, objName)
            if tfId:
                if not self._is_final(rc, tfId):
                    self._register(rc, workType, tfId, cnt)
                return

    # -- MODIFIED -----------------------------------------------------------
    def modifEvent(self, objName, eStatus):
        rc = self.rc
        tfId, seq, cnt = self.searchPattern("ps", objName)
        if tfId and self._is_final(rc, tfId):
            return
        if tfId:
            # a PS that exits by itself (signal from a worker, or a crash)
            if _get(eStatus, "failed"):
                self._register(rc, "ps", tfId, cnt)
                msg = _get(eStatus, "message")
                self._job_failed(rc, tfId, "ps task %s failed%s" % (objName, (": %s" % msg) if msg else ""))
            return
        tfId, seq, cnt = self.searchPattern("worker", objName)
        if not tfId or self._is_final(rc, tfId):
            return
        self._register(rc, "worker", tfId, cnt)
        if _get(eStatus, "succeeded") == 1:
            if not self._first_time(rc, "worker_done", tfId, seq):
                return
            curCount = int(rc.hincrby(self._key("worker_key"), tfId, -1))
            if curCount == 0:
                log.info("all workers of %s done, enqueue cleanup", tfId)
                self._enqueue_cleanup(rc, tfId, cnt, kill_workers=False)
            else:
                log.info("worker %s of %s done, %d remaining", seq, tfId, curCount)
        elif _get(eStatus, "failed"):
            msg = _get(eStatus, "message")          # the launcher's own verdict, e.g. DeadlineExceeded
            code = _get(eStatus, "exitCode")
            if not msg and code not in (None, 0, 1):
                msg = "exit code %s" % code         # 1 = an exception, reported by the task itself (_job_failed)
            self._job_failed(rc, tfId, "worker task %s failed%s" % (objName, (": %s" % msg) if msg else ""), worker_cnt=cnt)

    def _counts(self, rc, tfId, worker_cnt=None):
        psCnt = rc.hget(self._key("ps_key"), tfId)
        if psCnt is None:
            # ADDED for the PS side not seen yet: fall back to the job record
            raw = rc.get(tfId)
            try:
                psCnt = len(json.loads(raw)["ps"]) if raw else 0
            except (ValueError, KeyError):
                psCnt = 0
            # registered count == original count, remaining may be lower
        else:
            # hash holds the *remaining* count; original is in the record/seen set
            psCnt = int(psCnt) + int(rc.scard("ps_deleted:seen:%s" % tfId) or 0)
        if worker_cnt is None:
            raw = rc.get(tfId)
            try:
                worker_cnt = len(json.loads(raw)["worker"]) if raw else 0
            except (ValueError, KeyError):
                worker_cnt = 0
        return int(psCnt), int(worker_cnt)

    def _enqueue_cleanup(self, rc, tfId, worker_cnt, kill_workers, reason=None):
        psCnt, worker_cnt = self._counts(rc, tfId, worker_cnt)
        allPs = [task_name(tfId, "ps", i, psCnt) for i in range(psCnt)]
        allWorker = [task_name(tfId, "worker", i, worker_cnt) for i in range(worker_cnt)]
        tfInfo = {"ps": allPs, "worker": allWorker}
        if kill_workers:
            tfInfo["kill_workers"] = True
            tfInfo["reason"] = reason
        rc.rpush(self._key("delete_queue"), json.dumps(tfInfo))
        if not kill_workers:
            self._update_record(rc, tfId, status="finishing", t_workers_done=time.time())
        if psCnt == 0:
            # nothing will ever emit a PS DELETED event: finish right here
            self._finalize(rc, tfId)

    def _job_failed(self, rc, tfId, reason, worker_cnt=None):
        if not rc.set("failed:seen:%s" % tfId, reason, nx=True):
            return
        try:
            # the failing task reports its exception into <uid>:metrics before it exits (service/entry.py)
            errs = {k: v for k, v in (rc.hgetall("%s:metrics" % tfId) or {}).items() if str(k).startswith("error:")}
            if errs:
                k = sorted(errs)[0]
                reason = "%s (%s: %s)" % (reason, str(k)[6:], errs[k])
        except Exception:       # noqa: BLE001 -- the reason is a convenience, the failure handling is not
            pass
        log.warning("job %s failed: %s", tfId, reason)
        self._update_record(rc, tfId, status="failed", reason=reason, t_failed=time.time())
        self._enqueue_cleanup(rc, tfId, worker_cnt, kill_workers=True, reason=reason)

    # -- DELETED ------------------------------------------------------------
    def delEvent(self, objName, eStatus):
        rc = self.rc
        tfId, seq, cnt = self.searchPattern("ps", objName)
        if not tfId or self._is_final(rc, tfId):
            return
        self._register(rc, "ps", tfId, cnt)
        if not self._first_time(rc, "ps_deleted", tfId, seq):
            return
        psCurCount = int(rc.hincrby(self._key("ps_key"), tfId, -1))
        if psCurCount == 0:
            self._finalize(rc, tfId)

    def _finalize(self, rc, tfId):
        rc.hdel(self._key("ps_key"), tfId)
        rc.hdel(self._key("worker_key"), tfId)
        failed = rc.get("failed:seen:%s" % tfId) is not None
        rc.delete("worker_done:seen:%s" % tfId, "ps_deleted:seen:%s" % tfId, "failed:seen:%s" % tfId)
        cfg = self.cfg
        rc.srem(cfg.get("redis", "running_set"), tfId)
        if failed:
            rc.sadd(cfg.get("redis", "failed_set", "failed_uids"), tfId)
            self._update_record(rc, tfId, status="failed", t_cleaned=time.time())
        else:
            rc.sadd(cfg.get("redis", "success_set"), tfId)
            self._update_record(rc, tfId, status="succeeded", t_cleaned=time.time())
        log.info("job %s cleaned up (%s)", tfId, "failed" if failed else "succeeded")
