"""Lifecycle state machine against an in-memory store (SURVEY 4.1 'Unit -- state
machine'): in-order, out-of-order, duplicated events and failures."""
import json

import pytest

from distributed_tensorflow_amd.event_handlers import load_handler
from distributed_tensorflow_amd.event_handlers.CleanJobHandler import CleanJobHandler
from distributed_tensorflow_amd.event_handlers.EventHandler import EventHandler
from distributed_tensorflow_amd.utils.naming import task_name

UID = "3e8f3702-d10b-11e8-abe4-fa163ef8da8a"
OK = {"succeeded": 1}
RUN = {"active": 1}
FAIL = {"failed": 1}


def names(p, w):
    return ([task_name(UID, "ps", i, p) for i in range(p)], [task_name(UID, "worker", i, w) for i in range(w)])


def submit_record(store, p, w):
    ps, wk = names(p, w)
    store.sadd("running_uids", UID)
    store.set(UID, json.dumps({"ps": [n + ":2222" for n in ps], "worker": [n + ":2222" for n in wk],
                               "status": "running"}))
    return ps, wk


def test_handler_loading_by_config_name():
    assert load_handler("CleanJobHandler") is CleanJobHandler
    assert issubclass(CleanJobHandler, EventHandler)


def test_base_handler_dispatch():
    seen = []

    class H(EventHandler):
        def addEvent(self, n, s): seen.append(("A", n))
        def modifEvent(self, n, s): seen.append(("M", n))
        def delEvent(self, n, s): seen.append(("D", n))
        def errorEvent(self, n, s): seen.append(("E", n))

    h = H()
    for t in ("ADDED", "MODIFIED", "DELETED", "BOOKMARK"):
        h(t, "x", None)
    assert seen == [("A", "x"), ("M", "x"), ("D", "x"), ("E", "x")]


def test_happy_path_matches_reference_keys(store):
    h = CleanJobHandler()
    ps, wk = submit_record(store, 1, 2)
    for n in ps + wk:
        h("ADDED", n, RUN)
    assert store.hget("tf_ps_remaining", UID) == "1"       # HSETNX ps_key uid count
    assert store.hget("tf_worker_remaining", UID) == "2"
    h("MODIFIED", wk[0], OK)
    assert store.hget("tf_worker_remaining", UID) == "1"
    assert store.llen("tf_delete_queue") == 0
    h("MODIFIED", wk[1], OK)
    item = json.loads(store.lpop("tf_delete_queue"))
    assert item == {"ps": ps, "worker": wk}                 # names without port
    assert json.loads(store.get(UID))["status"] == "finishing"
    h("DELETED", ps[0], None)
    assert store.hget("tf_ps_remaining", UID) is None and store.hget("tf_worker_remaining", UID) is None
    rec = json.loads(store.get(UID))
    assert rec["status"] == "succeeded"
    assert store.smembers("running_uids") == set() and store.smembers("success_uids") == {UID}


def test_duplicate_and_replayed_events_do_not_double_decrement(store):
    h = CleanJobHandler()
    ps, wk = submit_record(store, 2, 2)
    for n in ps + wk + ps + wk:           # watch restart replays ADDED
        h("ADDED", n, RUN)
    assert store.hget("tf_worker_remaining", UID) == "2"
    for _ in range(3):                    # several MODIFIED with succeeded == 1 for one job
        h("MODIFIED", wk[0], OK)
    assert store.hget("tf_worker_remaining", UID) == "1"
    assert store.llen("tf_delete_queue") == 0
    h("MODIFIED", wk[1], OK)
    h("MODIFIED", wk[1], OK)
    assert store.llen("tf_delete_queue") == 1
    h("DELETED", ps[0], None)
    h("DELETED", ps[0], None)
    assert store.hget("tf_ps_remaining", UID) == "1"
    h("DELETED", ps[1], None)
    assert json.loads(store.get(UID))["status"] == "succeeded"


def test_out_of_order_modified_before_added(store):
    """The reference's 4-process pool can run HINCRBY before HSETNX, leaving -1."""
    h = CleanJobHandler()
    ps, wk = submit_record(store, 1, 2)
    h("MODIFIED", wk[0], OK)              # before any ADDED
    assert store.hget("tf_worker_remaining", UID) == "1"
    h("ADDED", wk[0], RUN)
    h("ADDED", wk[1], RUN)
    h("MODIFIED", wk[1], OK)
    item = json.loads(store.lpop("tf_delete_queue"))
    assert item["ps"] == ps               # PS count recovered from the job record
    h("ADDED", ps[0], RUN)
    h("DELETED", ps[0], None)
    assert json.loads(store.get(UID))["status"] == "succeeded"


def test_failed_worker_reaps_whole_job(store):
    h = CleanJobHandler()
    ps, wk = submit_record(store, 1, 2)
    for n in ps + wk:
        h("ADDED", n, RUN)
    h("MODIFIED", wk[1], FAIL)
    h("MODIFIED", wk[1], FAIL)            # idempotent
    assert store.llen("tf_delete_queue") == 1
    item = json.loads(store.lpop("tf_delete_queue"))
    assert item["kill_workers"] is True and item["ps"] == ps and item["worker"] == wk
    assert json.loads(store.get(UID))["status"] == "failed"
    h("DELETED", wk[0], None)
    h("DELETED", ps[0], None)
    rec = json.loads(store.get(UID))
    assert rec["status"] == "failed" and "t_cleaned" in rec
    assert store.smembers("failed_uids") == {UID} and store.smembers("running_uids") == set()
    # late events after cleanup must not resurrect state
    h("MODIFIED", wk[0], FAIL)
    h("ADDED", ps[0], RUN)
    assert store.llen("tf_delete_queue") == 0 and store.hget("tf_ps_remaining", UID) is None


def test_job_without_ps_finishes_at_last_worker(store):
    h = CleanJobHandler()
    _, wk = submit_record(store, 0, 1)
    h("ADDED", wk[0], RUN)
    h("MODIFIED", wk[0], OK)
    assert json.loads(store.get(UID))["status"] == "succeeded"
    assert json.loads(store.lpop("tf_delete_queue")) == {"ps": [], "worker": wk}


def test_foreign_job_names_are_ignored(store):
    h = CleanJobHandler()
    h("ADDED", "some-other-job", RUN)
    h("MODIFIED", "tf-not-a-uuid-worker-0-1", OK)
    assert store.keys("*") == []
