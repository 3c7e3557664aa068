"""Warm rank process ("warm pre-forked rank pool").

Half of submit -> first-step latency was the interpreter start + ``import torch`` of every task process
(1.2 s of 2.33 s on the GPU box, round 1).  The launcher therefore keeps a few of these processes around:
each has imported torch, torch.distributed and this package's model / trainer modules -- but has NOT touched
HIP (no device call, the kernel library not loaded), so the task's ``HIP_VISIBLE_DEVICES`` still takes effect --
and blocks on stdin.  A job assignment is one JSON line ``{"argv": [...], "env": {...}, "log": path}``: the
process adopts the environment, points stdout / stderr at the task's log file and becomes that task
(``service.entry.main(argv)``); its exit code is the task's.  EOF on stdin (launcher gone) = exit.
"""
from __future__ import annotations

import json
import os
import sys


def preload():
    from ..utils import miopen as miopen_cache

    miopen_cache.setup()      # the shipped MIOpen find-db / kernel cache must be in the environment before torch loads
    import torch  # noqa: F401
    import torch.distributed  # noqa: F401

    from ..models import get_model_module  # noqa: F401
    from ..parallel import async_ps, checkpoint, dedicated_ps, ps_runtime  # noqa: F401
    from ..train import jobspec, trainer  # noqa: F401
    from . import entry  # noqa: F401

    return entry


def main():
    entry = preload()
    sys.stdout.write("WARM %d\n" % os.getpid())
    sys.stdout.flush()
    line = sys.stdin.readline()
    if not line.strip():
        return 0
    job = json.loads(line)
    os.environ.update({str(k): str(v) for k, v in job.get("env", {}).items()})
    for k in job.get("unset", []):
        os.environ.pop(k, None)
    log = job.get("log")
    if log:
        fd = os.open(log, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
        sys.stdout.flush()
        os.dup2(fd, 1)
        os.dup2(fd, 2)
        os.close(fd)
    dn = os.open(os.devnull, os.O_RDONLY)
    os.dup2(dn, 0)
    os.close(dn)
    # a freshly started interpreter would have re-read these at import time
    from ..utils.config import ApiConfig

    try:
        ApiConfig.reload()
    except AttributeError:
        pass
    # a private MIOpen db / kernel-cache copy per task, like a cold start (preload used a shared placeholder)
    os.environ.pop("MIOPEN_USER_DB_PATH", None)
    os.environ.pop("MIOPEN_CUSTOM_CACHE_DIR", None)
    print("[warm] pid %d adopted task %s" % (os.getpid(), job.get("env", {}).get("DTF_TASK_NAME", "?")), flush=True)
    rc = entry.main([str(a) for a in job["argv"]])
    sys.stdout.flush()
    return int(rc or 0)


if __name__ == "__main__":
    from .entry import leave

    try:
        _rc = main()
    except SystemExit as e:
        _rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    except BaseException:           # noqa: BLE001
        import traceback

        traceback.print_exc()
        _rc = 1
    leave(_rc)
