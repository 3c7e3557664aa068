"""List or print what a checkpoint directory holds, or gather it into one file -- what TF1's ``inspect_checkpoint`` /
``tf.train.load_checkpoint`` do for the checkpoints the reference's jobs leave in ``export``.

    python -m distributed_tensorflow_amd.tools.inspect_checkpoint /tmp/out                       # names, shapes, dtypes
    python -m distributed_tensorflow_amd.tools.inspect_checkpoint /tmp/out --tensor_name hid_w   # one variable
    python -m distributed_tensorflow_amd.tools.inspect_checkpoint /tmp/out --all_tensors
    python -m distributed_tensorflow_amd.tools.inspect_checkpoint /tmp/out --step 800            # an older checkpoint
    python -m distributed_tensorflow_amd.tools.inspect_checkpoint /tmp/out --export model.safetensors   # or .pt

A checkpoint is sharded over the ranks that wrote it (``parallel/checkpoint.py``); the reader reassembles variables
whatever that number was, so this runs anywhere, without the job's world size and without a GPU."""
from __future__ import annotations

import argparse
import json
import sys

import torch

from ..parallel import checkpoint as ckpt


def summarize(export_dir, step=None):
    rd = ckpt.CheckpointReader(export_dir, step)
    rows = []
    for n in rd.names():
        v = rd.index["variables"][n]
        rows.append({"name": n, "shape": list(v["shape"]), "dtype": v.get("dtype", "float32"),
                     "elements": int(torch.Size(v["shape"]).numel()) if v["shape"] else 1,
                     "pieces": len(v["pieces"])})
    return rd, rows


def gather(rd, slots=False):
    """``{name: tensor}`` of the model variables (optimizer slots too with ``slots``) plus the replicated state."""
    opt_slots = ("Momentum", "Adam", "Adam_1")
    out = {}
    for n in rd.names():
        if not slots and n.rsplit("/", 1)[-1] in opt_slots:
            continue
        out[n] = rd.read(n).contiguous()
    for n, t in rd.replicated().items():
        out.setdefault(n, t if isinstance(t, torch.Tensor) else torch.as_tensor(t))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(prog="inspect_checkpoint", description=__doc__.split("\n\n")[0])
    ap.add_argument("export_dir")
    ap.add_argument("--step", type=int, default=None, help="default: the newest complete checkpoint")
    ap.add_argument("--tensor_name", default=None)
    ap.add_argument("--all_tensors", action="store_true")
    ap.add_argument("--json", action="store_true", help="machine-readable listing")
    ap.add_argument("--export", default=None, help="write the variables into ONE file (.safetensors or .pt)")
    ap.add_argument("--with_slots", action="store_true", help="--export: optimizer slots too")
    a = ap.parse_args(argv)
    try:
        rd, rows = summarize(a.export_dir, a.step)
    except FileNotFoundError as e:
        print(str(e), file=sys.stderr)
        return 1
    if a.export:
        tensors = gather(rd, a.with_slots)
        if a.export.endswith(".safetensors"):
            from safetensors.torch import save_file

            save_file({k: v.contiguous() for k, v in tensors.items()}, a.export,
                      metadata={"step": str(rd.step), "format": "dtf-ckpt-1"})
        else:
            torch.save(tensors, a.export)
        print("wrote %d tensors of step %d to %s" % (len(tensors), rd.step, a.export))
        return 0
    if a.tensor_name:
        if not rd.has(a.tensor_name):
            print("no variable %r in the checkpoint of step %d (try without --tensor_name for the list)" % (
                a.tensor_name, rd.step), file=sys.stderr)
            return 1
        t = rd.read(a.tensor_name)
        print("tensor_name: %s  shape %s  dtype %s" % (a.tensor_name, list(t.shape), str(t.dtype).replace("torch.", "")))
        torch.set_printoptions(precision=6, threshold=2000, linewidth=120)
        print(t)
        return 0
    if a.json:
        print(json.dumps({"step": rd.step, "world": rd.world, "optimizer": rd.index.get("optimizer"), "variables": rows}))
        return 0
    total = sum(r["elements"] for r in rows)
    print("checkpoint %s, step %d, written by %d rank(s), optimizer %s, steps kept %s" % (
        a.export_dir, rd.step, rd.world, rd.index.get("optimizer", "?"), ckpt.checkpoint_steps(a.export_dir)))
    w = max([len(r["name"]) for r in rows] + [4])
    for r in rows:
        print("%-*s  %-10s %-22s %d" % (w, r["name"], r["dtype"], r["shape"], r["elements"]))
        if a.all_tensors:
            print(rd.read(r["name"]))
    rep = rd.replicated()
    for n in sorted(rep):
        t = rep[n]
        print("%-*s  %-10s %-22s (replicated)" % (w, n, str(getattr(t, "dtype", type(t).__name__)).replace("torch.", ""),
                                                  list(getattr(t, "shape", []))))
    print("%d variables, %d elements" % (len(rows), total))
    return 0


if __name__ == "__main__":
    sys.exit(main())
