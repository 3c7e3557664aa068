"""Task naming shared by the submitter, the watcher and the reaper.

``tfId = "tf-" + uid + "-" + workType + "-" + seq + "-" + count`` (reference
``handlers/train_handler.py:23,61``); the total ``count`` is embedded so the
watcher can recover it from the name alone (``eventHandlers/CleanJobHandler.py:
10-17``).  The reference's regexes use ``([0-9].*)`` groups, which mis-split
names with multi-digit fields greedily; the patterns here are anchored and
digit-only.
"""
from __future__ import annotations

import re

UUID_RE = r"[0-9a-z]{8}-[0-9a-z]{4}-[0-9a-z]{4}-[0-9a-z]{4}-[0-9a-z]{12}"
_TASK_RE = re.compile(r"^tf-(" + UUID_RE + r")-([a-z][a-z0-9_]*)-([0-9]+)-([0-9]+)$")

TASK_TYPES = ("ps", "worker")


def task_name(uid, work_type, seq, count):
    return "-".join(["tf", str(uid), str(work_type), str(seq), str(count)])


def parse_task_name(name):
    """``name`` -> ``(uid, work_type, seq, count)`` or ``None``."""
    m = _TASK_RE.match(name or "")
    if not m:
        return None
    return m.group(1), m.group(2), int(m.group(3)), int(m.group(4))


def search_pattern(work_type, name):
    """Reference-shaped helper: ``(uid, seq, count)`` strings or ``(None,)*3``
    (``eventHandlers/CleanJobHandler.py:13-17``)."""
    p = parse_task_name(name)
    if p is None or p[1] != work_type:
        return None, None, None
    return p[0], str(p[2]), str(p[3])


def host_list(uid, work_type, count, port):
    return ["%s:%s" % (task_name(uid, work_type, i, count), port) for i in range(int(count))]
