"""Ship-with-the-tree MIOpen tuning results.

PyTorch-ROCm runs the convolutions of the conv nets through MIOpen.  On a fresh box MIOpen
(a) compiles every kernel it picks (~60 s for ResNet-50's first steps) and (b) picks solvers
from heuristics unless an exhaustive ``find`` has been run (~200 s for ResNet-50, +12.6 %
images/s on MI355X).  Both are paid once, offline (``scripts/gpu_miopen_tune.sh``); the
resulting user find-db / perf-db (text) and kernel cache (``*.ukdb``) live in
``distributed_tensorflow_amd/miopen_db`` and :func:`setup` points MIOpen at a private
writable copy of them.  This is what keeps job submit -> first step short (SURVEY 7.2
item 2: never JIT or tune at submit time) without giving up the tuned kernels.

Shapes that are NOT in the shipped databases (another batch size, another model) used to cost 13 s (ResNet-50) to
36 s (Inception-v3) before the first step: MIOpen's heuristics picked Winograd / direct / GEMM solvers whose HIP and
OpenCL sources it compiles at run time.  :func:`restrict_solvers` (called by :func:`setup`) leaves only the implicit-GEMM
family enabled: composable_kernel instances compiled into the library plus MIOpen's shape-independent gfx950 assembly
kernels, all of which the shipped kernel cache already holds (``scripts/gpu_miopen_widen.sh``).  Measured on MI355X,
empty caches, batch 64 (``profiles/r2_miopen_cache_miss.txt``): first step 12.8 -> 2.6 s and 34.6 -> 5.0 s, steady state
+9 % / +20 % (the heuristics' non-igemm picks were also slower); neutral for the tuned batch-256 shapes, whose find-db
entries are implicit-GEMM solvers anyway.
"""
from __future__ import annotations

import os
import shutil
import tempfile

_SRC = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "miopen_db"))


_NON_IGEMM = ("MIOPEN_DEBUG_CONV_DIRECT", "MIOPEN_DEBUG_CONV_WINOGRAD", "MIOPEN_DEBUG_CONV_FFT", "MIOPEN_DEBUG_CONV_GEMM")


def restrict_solvers():
    """Disable the solver families that compile HIP / OpenCL at run time (see module docstring).  A value the user
    exported wins; ``DTF_MIOPEN_ALL_SOLVERS=1`` leaves MIOpen's defaults alone."""
    if os.environ.get("DTF_MIOPEN_ALL_SOLVERS", "0") == "1":
        return False
    mine = []
    for k in _NON_IGEMM:
        if k not in os.environ:
            os.environ[k] = "0"
            mine.append(k)
    if mine:
        # remembered so that unrestrict() removes exactly what this function added, never a value the user exported
        os.environ["DTF_MIOPEN_RESTRICTED"] = ",".join(
            sorted(set(mine) | set(filter(None, os.environ.get("DTF_MIOPEN_RESTRICTED", "").split(",")))))
    return True


def unrestrict():
    """Undo :func:`restrict_solvers` for code that is not this framework's: a user ``file=`` script gets MIOpen with
    its own defaults (grouped / depthwise / 3-D / odd fp32 shapes may have no implicit-GEMM solver at all), whether
    the process is a cold start or a warm rank that had called :func:`setup` while idle.  Effective as
    long as MIOpen has not created a handle yet, which is the case in both."""
    for k in filter(None, os.environ.pop("DTF_MIOPEN_RESTRICTED", "").split(",")):
        os.environ.pop(k, None)


def setup(force=False):
    """Call before the first convolution (ideally before ``import torch``).  Idempotent.
    Respects MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR if the user set them."""
    restrict_solvers()
    if not force and ("MIOPEN_USER_DB_PATH" in os.environ or "MIOPEN_CUSTOM_CACHE_DIR" in os.environ):
        return os.environ.get("MIOPEN_USER_DB_PATH")
    if not os.path.isdir(_SRC) or not os.listdir(_SRC):
        return None
    # one private copy per local rank: eight processes of one node otherwise share (and lock) one sqlite
    # kernel cache and one find-db file while they all start at once
    # DTF_MIOPEN_TAG: a GPU-hot warm worker (service/warm_worker.py) creates its MIOpen handle BEFORE it knows which
    # task it will become, so its copy is keyed by its own pid instead of by a rank
    rank = os.environ.get("DTF_MIOPEN_TAG") or (os.environ.get("DTF_JOB_NAME", "") + os.environ.get(
        "DTF_TASK_INDEX", os.environ.get("RANK", os.environ.get("LOCAL_RANK", "0"))))
    base = os.path.join(os.environ.get("TMPDIR", tempfile.gettempdir()), "dtf_miopen_%d_r%s" % (os.getuid(), rank))
    db, cache = os.path.join(base, "db"), os.path.join(base, "cache")
    try:
        os.makedirs(db, exist_ok=True)
        os.makedirs(cache, exist_ok=True)
        for f in os.listdir(_SRC):
            dst = os.path.join(cache if f.endswith(".ukdb") else db, f)
            if not os.path.exists(dst):
                tmp = dst + ".tmp.%d" % os.getpid()
                shutil.copyfile(os.path.join(_SRC, f), tmp)
                os.replace(tmp, dst)
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = db
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = cache
    return db


def cleanup_tagged():
    """Remove the private copy of a ``DTF_MIOPEN_TAG`` process (called when that process ends)."""
    tag = os.environ.get("DTF_MIOPEN_TAG")
    if tag:
        base = os.path.join(os.environ.get("TMPDIR", tempfile.gettempdir()), "dtf_miopen_%d_r%s" % (os.getuid(), tag))
        shutil.rmtree(base, ignore_errors=True)
