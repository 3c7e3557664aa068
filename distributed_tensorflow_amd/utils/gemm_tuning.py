"""Ship-with-the-tree GEMM solution selection for the plain (library) GEMMs.

The 1x1-convolution data gradients of the conv nets, BERT's QKV / dX projections and the classifier heads run through
hipBLASLt / rocBLAS (``torch.matmul``).  Their default heuristics are not the fastest solution for several of these
shapes; PyTorch's TunableOp can time every solution per shape and record the winner.  That tuning is paid once,
offline (``scripts/gpu_tunableop.sh``, 35-50 s per model on MI355X); the winners live in
``distributed_tensorflow_amd/gemm_db/gfx950_tunableop.csv`` and :func:`setup` makes every task READ them (tuning
itself stays off: a shape that is not in the file takes the library default, nothing is timed at run time).
Measured (``profiles/r2_tunableop.txt``): ResNet-50 +1.0 %, BERT-base +1.4 %, Inception-v3 +-0.
Cost: TunableOp resolves the recorded solution names against the libraries at the first GEMM of each kind, 1.0-1.5 s
before the first step.  ``bench.py`` (steady-state throughput) always enables it; a job does from 10 000 steps
(``gemm_tuning=on|off|auto`` in the job's ``file`` options, ``service/entry.py``).

The file starts with validator lines (PyTorch / HIP / hipBLASLt / rocBLAS versions, GPU architecture): TunableOp
ignores the whole file when they do not match the running stack, so a newer ROCm degrades to the defaults.
"""
from __future__ import annotations

import atexit
import os
import shutil
import tempfile

# DTF_GEMM_DB: another solution file (A/B of a re-tuned file, scripts/gpu_retune.sh)
_SRC = os.environ.get("DTF_GEMM_DB") or os.path.normpath(
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gemm_db", "gfx950_tunableop.csv"))
_MAX_DEVICES = 16


_active = False


def active():
    """True when :func:`setup` pointed TunableOp at the shipped file in this process."""
    return _active and os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1"


_fwd_shapes = None


def deactivate():
    """The process stops using the shipped choices (``tunable.enable(False)`` is the caller's)."""
    global _active, _fwd_shapes
    _active, _fwd_shapes = False, None


def _validator_ok(key, value):
    """The two validators that can be checked from Python without touching the libraries (the others are left to
    TunableOp itself)."""
    try:
        import torch

        if key == "PT_VERSION":
            return torch.__version__.split("+")[0].startswith(value)
        if key == "GCN_ARCH_NAME" and torch.cuda.is_available():
            return torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName == value
    except Exception:
        return True
    return True


def tuned_gemm_shapes():
    """``{(op, m, n, k)}`` of the shipped file, op in tn / nn / nt, (m, n, k) in the library's column-major naming:
    for ``y[rows, cout] = x[rows, cin] . w[cout, cin]^T`` that is ``("tn", cout, rows, cin)``, for
    ``dx[rows, cin] = dy[rows, cout] . w[cout, cin]`` it is ``("nn", cin, rows, cout)``.  Empty when the shipped
    choices are not in use: ``ops/conv.py`` sends a 1x1 convolution product to the library only when its exact GEMM
    has a recorded solution (with the default heuristics MIOpen is faster below 1024 input channels: 10 700 vs 10 490
    images/s; with the recorded solutions the library wins: 10 790 -> 11 150, ``profiles/r2_tunableop.txt``)."""
    global _fwd_shapes
    if not active():
        return frozenset()
    if _fwd_shapes is None:
        out = set()
        try:
            for line in open(_SRC):
                parts = line.strip().split(",")
                if parts[0] == "Validator" and len(parts) >= 3 and not _validator_ok(parts[1], parts[2]):
                    out = set()             # TunableOp will ignore the file: do not route anything by it either
                    break
                if len(parts) < 3 or not parts[0].startswith("GemmTunableOp_BFloat16_"):
                    continue
                f = parts[1].split("_")
                if len(f) >= 4 and f[0] in ("tn", "nn", "nt"):
                    out.add((f[0], int(f[1]), int(f[2]), int(f[3])))
        except (OSError, ValueError):
            pass
        _fwd_shapes = frozenset(out)
    return _fwd_shapes


def after_torch_import():
    """Read-only use: never write the file back when the process ends."""
    if not active():
        return
    try:
        import torch.cuda.tunable as tunable

        tunable.write_file_on_exit(False)
    except Exception:       # older / newer torch without the switch: the private copy makes the write harmless
        pass


def setup():
    """Call before the first GEMM.  Idempotent; any PYTORCH_TUNABLEOP_* variable the user exported wins;
    ``DTF_GEMM_TUNING=0`` switches it off."""
    if os.environ.get("DTF_GEMM_TUNING", "1") == "0" or not os.path.isfile(_SRC):
        return None
    if any(k.startswith("PYTORCH_TUNABLEOP_") for k in os.environ):
        return os.environ.get("PYTORCH_TUNABLEOP_FILENAME")
    # a PRIVATE copy per process: TunableOp may rewrite its file when the process ends, and tasks of different jobs all
    # see "device 0" behind HIP_VISIBLE_DEVICES -- a shared file could be rewritten by two of them at once
    base = os.path.join(os.environ.get("TMPDIR", tempfile.gettempdir()), "dtf_gemm_%d" % os.getuid(), str(os.getpid()))
    try:
        os.makedirs(base, exist_ok=True)
        # TunableOp inserts the device ordinal in front of the extension: one copy per ordinal a rank may use
        for d in range(_MAX_DEVICES):
            shutil.copyfile(_SRC, os.path.join(base, "gemm%d.csv" % d))
    except OSError:
        return None
    atexit.register(shutil.rmtree, base, True)
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
    os.environ["PYTORCH_TUNABLEOP_VERBOSE"] = "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = os.path.join(base, "gemm.csv")
    global _active
    _active = True
    return os.environ["PYTORCH_TUNABLEOP_FILENAME"]
