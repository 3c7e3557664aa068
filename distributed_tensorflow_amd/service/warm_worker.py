"""Warm rank process ("warm pre-forked rank pool").

Half of submit -> first-step latency was the interpreter start + ``import torch`` of every task process
(1.2 s of 2.33 s on the GPU box, round 1).  The launcher therefore keeps a few of these processes around:
each has imported torch, torch.distributed and this package's model / trainer modules -- but has NOT touched
HIP (no device call, the kernel library not loaded), so the task's ``HIP_VISIBLE_DEVICES`` still takes effect --
and blocks on stdin.

GPU-HOT variant (``DTF_WARM_HOT=1``, one per free GPU, started by the launcher with ``HIP_VISIBLE_DEVICES``
already pinned to that GPU): everything above PLUS the HIP context, the in-tree kernel library with its code
object loaded, a MIOpen handle with the shipped find-db / kernel cache open, hipBLASLt with the shipped GEMM
solution file resolved (the 1.0-1.5 s TunableOp look-up that cold REST jobs under 10 000 steps skip), the
allocator warmed, and optionally (``DTF_WARM_MODEL="resnet50?batch=256"``) one forward/backward of the model
family the node mostly runs, so that MIOpen's per-layer kernels are already loaded in the process.  The launcher
hands a job to the hot worker of the GPU it scheduled; the worker BECOMES the task (no exec, no re-launch of a
GPU-initialised process).  Round 2 measured 1.40 s submit -> step 1 with the HIP-free pool (1.3 s of it HIP
context + library loads + first step); this is what removes that part.

A job assignment is one JSON line ``{"argv": [...], "env": {...}, "log": path}``: the
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


def hot_preload():
    """Everything a GPU task pays before its first step that does not depend on the job (see module docstring).
    Returns a dict of what was warmed (goes into the task log for the submit -> step-1 record)."""
    import time

    t0 = time.time()
    done = {}
    import torch

    from ..utils import gemm_tuning

    gemm_tuning.after_torch_import()
    if not torch.cuda.is_available():
        return {"skipped": "no device visible"}         # CPU box (tests): a pinned, HIP-free warm rank
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x = torch.zeros(1 << 22, device=dev)              # HIP context + allocator
    torch.cuda.synchronize()
    done["hip_context_s"] = round(time.time() - t0, 3)
    t1 = time.time()
    from ..ops import hip, optim as optim_ops

    hip.lib()
    n = 4096                                          # one own kernel: loads the library's code object
    g = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    p32, m = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    optim_ops.sgd_momentum_apply(g, p32, m, torch.zeros(n, dtype=torch.bfloat16, device=dev), lr=0.1, momentum=0.9)
    torch.cuda.synchronize()
    done["kernel_library_s"] = round(time.time() - t1, 3)
    t1 = time.time()
    a = torch.randn(256, 256, device=dev, dtype=torch.bfloat16)
    (a @ a).sum().item()                              # hipBLASLt handle + the shipped solution file resolved
    done["gemm_library_s"] = round(time.time() - t1, 3)
    t1 = time.time()
    conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev, torch.bfloat16).to(memory_format=torch.channels_last)
    xi = torch.randn(8, 64, 14, 14, device=dev, dtype=torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_()
    conv(xi).sum().backward()                         # MIOpen handle, find-db and kernel cache open
    torch.cuda.synchronize()
    done["miopen_s"] = round(time.time() - t1, 3)
    hint = os.environ.get("DTF_WARM_MODEL", "")
    if hint:
        t1 = time.time()
        try:
            from ..models import get_model_module
            from ..parallel.comm import Comm
            from ..parallel.ps_runtime import ShardedPS
            from ..train.jobspec import parse_file_field

            _, opts = parse_file_field("builtin://" + hint)
            mod = get_model_module(opts.model)
            model = mod.build(**opts.extra)
            model = mod.prepare(model, dev) if hasattr(mod, "prepare") else model.to(dev)
            ps = ShardedPS(list(model.named_parameters()), Comm.single(dev), mod.default_optimizer())
            bkw = {k: int(v) for k, v in opts.extra.items() if k in ("seq_len", "image_size")}
            batch = mod.synthetic_batch(opts.batch, device=dev, seed=1, **bkw)
            for _ in range(2):
                mod.loss_fn(model, batch).backward()
                ps.step()
            torch.cuda.synchronize()
            ps.remove_hooks()
            del ps, model, batch
            done["model_hint"] = hint
        except Exception as e:      # noqa: BLE001 -- a bad hint only means a colder first step
            done["model_hint_error"] = "%s: %s" % (type(e).__name__, e)
        done["model_hint_s"] = round(time.time() - t1, 3)
    del x, g, p32, m, a, conv, xi
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    done["total_s"] = round(time.time() - t0, 3)
    return done


def main():
    hot = os.environ.get("DTF_WARM_HOT", "0") == "1"
    warmed = None
    if hot:
        os.environ["DTF_MIOPEN_TAG"] = "hot%d" % os.getpid()
        from ..utils import gemm_tuning

        gemm_tuning.setup()       # before torch loads: every job on a hot worker runs on the shipped GEMM choices
    entry = preload()
    if hot:
        warmed = hot_preload()
    sys.stdout.write("WARM %d%s\n" % (os.getpid(), " HOT" if hot else ""))
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
    if not hot:
        # a private MIOpen db / kernel-cache copy per task, like a cold start (preload used a shared placeholder).
        # A hot worker's MIOpen handle is already open on its own pid-keyed copy: it keeps it.
        os.environ.pop("MIOPEN_USER_DB_PATH", None)
        os.environ.pop("MIOPEN_CUSTOM_CACHE_DIR", None)
    print("[warm] pid %d adopted task %s%s" % (os.getpid(), job.get("env", {}).get("DTF_TASK_NAME", "?"),
                                               (" (GPU-hot: %s)" % json.dumps(warmed)) if hot else ""), flush=True)
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
