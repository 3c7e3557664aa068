"""Interpretation of the job spec's ``file`` / ``data`` / ``export`` fields.

The reference forwards them verbatim to ``/notebooks/entry.sh`` as argv[5..7]
(``handlers/train_handler.py:82-83``) where a script inside the opaque image
fetches ``file`` from HDFS and runs it.  Here ``file`` is either

* a built-in workload: ``resnet50`` or ``builtin://resnet50?steps=200&batch=256``
  (query keys override the ``[train]`` config section), or
* a path to a user python script, run with the cluster description in the
  environment (``DTF_JOB_NAME``, ``DTF_TASK_INDEX``, ``DTF_PS_HOSTS``,
  ``DTF_WORKER_HOSTS``, ``TF_CONFIG``, ``RANK``/``WORLD_SIZE``/``MASTER_ADDR``/``MASTER_PORT``)
  and the same nine arguments in ``sys.argv``.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from urllib.parse import parse_qsl, urlparse

from ..models import model_names
from ..utils.config import ApiConfig


@dataclass
class TrainOptions:
    model: str = "mnist_mlp"
    steps: int = 100
    batch: int = 64                 # per worker
    update_mode: str = "sync"       # sync | stale | async
    staleness: int = 1              # stale mode: updates become visible this many steps late
    ps_shards: str = "spec"         # spec: detail.ps shard groups (ps >= workers: one per worker) | auto: one per worker
    placement: str = "auto"         # auto | colocated | dedicated
    bucket_mb: float = 32.0
    wire_dtype: str = "auto"        # auto | bf16 | fp32
    push: str = "reduce_scatter"    # reduce_scatter | all_to_all
    lr: float = -1.0                # <0: model default
    lr_warmup_steps: int = 0        # linear warm-up of the learning rate over the first N steps of the JOB (a resumed job continues it)
    lr_schedule: str = "constant"   # after the warm-up: constant | cosine (to lr_final_fraction * lr at the last step) |
                                    # piecewise (lr *= lr_decay_factor at each of lr_boundaries, fractions of `steps`)
    lr_boundaries: str = "0.5,0.75"
    lr_decay_factor: float = 0.1
    lr_final_fraction: float = 0.0
    optimizer: str = ""             # "": model default
    ckpt_every: int = 0             # 0: only at the end (when export is set)
    keep_checkpoints: int = 5       # complete checkpoints kept in `export` (TF1 Saver's max_to_keep); 0: all
    resume: bool = False
    seed: int = 1234
    log_every: int = 10
    fault: str = ""                 # e.g. "exit:worker=1,step=3,code=17" / "hang:worker=0,step=2"
    graph: object = "default"       # 1: capture the whole training step in one hipGraph after 3 eager steps; "auto": capture,
                                    # then time eager steps against replays and keep the faster (small batches are
                                    # host-bound eagerly: ResNet-50 batch 32 runs 2.1x faster replayed, batch 256 does not);
                                    # 0: never.  "default" = auto for a one-worker GPU job (measured on one device), 0 for
                                    # jobs of several workers (a capture with live collectives is not measured there yet)
    gemm_tuning: str = "auto"       # shipped per-shape GEMM choices: on | off | auto (on from 10 000 steps: reading them
                                    # costs 1-1.5 s before the first step and returns 1-1.5 % per step)
    deterministic: int = -1         # -1: as the process was started (DTF_DETERMINISTIC, default 1) | 0 | 1 | 2 (strict: bitwise
                                    # reproducible at every batch size, ~2.5 % slower; docs/SWITCHES.md)
    input: str = "auto"             # auto: the job's `data` directory if it holds a dataset (train/data.py) | synthetic
    synthetic_batches: int = 1      # synthetic input: this many different batches, served round-robin
    extra: dict = field(default_factory=dict)


_INT = {"steps", "batch", "ckpt_every", "seed", "log_every", "staleness", "synthetic_batches", "keep_checkpoints",
        "deterministic", "lr_warmup_steps"}
_FLOAT = {"bucket_mb", "lr", "lr_decay_factor", "lr_final_fraction"}
_BOOL = {"resume", "graph"}


def _coerce(key, val):
    if key in _INT:
        return int(val)
    if key in _FLOAT:
        return float(val)
    if key == "graph" and str(val).lower() in ("auto", "default"):
        return str(val).lower()
    if key in _BOOL:
        return str(val).lower() in ("1", "true", "yes", "on")
    return val


def parse_file_field(file_field, cfg=None):
    """-> ``("builtin", TrainOptions)`` or ``("script", path)``; raises
    ``ValueError`` for something that is neither."""
    cfg = cfg or ApiConfig()
    opts = TrainOptions()
    for key in ("steps", "bucket_mb", "update_mode", "placement", "wire_dtype", "staleness", "ps_shards", "graph",
                "gemm_tuning"):
        if cfg.has_option("train", key):
            setattr(opts, key, _coerce(key, cfg.get("train", key)))
    f = (file_field or "").strip()
    u = urlparse(f)
    name = None
    if u.scheme == "builtin":
        name = (u.netloc + u.path).strip("/")
    elif u.scheme in ("", "file") and not os.path.exists(u.path):
        cand = u.path.strip("/").split("?")[0]
        if cand.lower().replace("-", "_") in model_names():
            name = cand
    if name is not None:
        opts.model = name.lower().replace("-", "_")
        for k, v in parse_qsl(u.query):
            if hasattr(opts, k) and k != "extra":
                setattr(opts, k, _coerce(k, v))
            else:
                opts.extra[k] = v
        return "builtin", opts
    path = u.path if u.scheme in ("", "file") else f
    if os.path.isfile(path):
        return "script", path
    raise ValueError("job 'file' %r is neither a built-in model (%s) nor an existing script"
                     % (file_field, ", ".join(model_names())))
