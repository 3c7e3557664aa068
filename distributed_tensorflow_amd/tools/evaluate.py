"""Evaluate what a training job left in its ``export`` directory on a dataset directory -- the evaluation job a TF1 user
runs next to the training job (restore the newest checkpoint, one pass over held-out data, loss and accuracy).

    python -m distributed_tensorflow_amd.tools.evaluate --model mnist_mlp --export /tmp/out --data /data/mnist-test
    python -m distributed_tensorflow_amd.tools.evaluate --model resnet50 --export /tmp/out --data /data/val --batch 256 \\
           [--step 800] [--device cuda:0] [--max_batches 100] [--option depth=101]

The checkpoint may come from any number of workers; the model runs in eval mode (BatchNorm uses the moving statistics the
chief saved) on one device.  Prints one JSON line: step, samples, loss, and accuracy for classification models."""
from __future__ import annotations

import argparse
import json
import sys
import types

import torch

from ..models import get_model_module
from ..parallel.comm import Comm
from ..parallel.ps_runtime import OptimizerSpec, ShardedPS
from ..train import data as D
from ..train.trainer import prepare_model, restore_checkpoint


def evaluate(model_name, export_dir, data_dir, batch=256, device=None, step=None, max_batches=None, options=None):
    mod = get_model_module(model_name)
    dev = torch.device(device) if device else torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    comm = Comm(0, 1, dev)
    model = prepare_model(mod, types.SimpleNamespace(extra=dict(options or {})), dev, comm)
    # the restore goes through the PS layout (it knows TF-style names, split variables, storage orders, table shards);
    # plain SGD = no optimizer slots to allocate
    ps = ShardedPS(list(model.named_parameters()), comm, OptimizerSpec(kind="sgd"), broadcast_init=False)
    tf_names = mod.tf_variable_names(model) if hasattr(mod, "tf_variable_names") else {}
    got = restore_checkpoint(model, ps, export_dir, tf_names, step=step)
    ds = D.open_dataset(data_dir, split="test")       # a directory with train-* and t10k-* files: the held-out ones
    if ds is None:
        raise FileNotFoundError("no dataset under %s (MNIST IDX files, images.npy + labels.npy, x.npy + y.npy or a manifest.json)" % data_dir)
    batch = min(int(batch), len(ds))
    stream = D.BatchStream(ds, batch, shuffle=False)
    n_batches = stream.per_epoch if not max_batches else min(stream.per_epoch, int(max_batches))
    batches = D.Prefetcher(stream, D.transform_for(mod, ds, dev), dev)
    model.eval()
    loss_sum = torch.zeros((), dtype=torch.float64, device=dev)
    hits = torch.zeros((), dtype=torch.int64, device=dev)
    counted = 0
    classify = ds.kind in ("mnist", "images")
    try:
        with torch.no_grad():
            for _ in range(n_batches):
                b = next(batches)
                loss_sum += mod.loss_fn(model, b).double()
                if classify:
                    logits = model(b[0])
                    if isinstance(logits, (tuple, list)):
                        logits = logits[0]
                    if logits.dim() == 2 and logits.shape[0] == b[1].shape[0]:
                        hits += (logits.float().argmax(-1) == b[1]).sum()
                        counted += int(b[1].shape[0])
    finally:
        batches.close()
    out = {"model": model_name, "step": int(got), "samples": n_batches * batch, "loss": float(loss_sum) / max(1, n_batches),
           "dataset": ds.source, "device": str(dev)}
    if counted:
        out["accuracy"] = float(hits) / counted
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(prog="evaluate", description=__doc__.split("\n\n")[0])
    ap.add_argument("--model", required=True)
    ap.add_argument("--export", required=True, help="the training job's export directory")
    ap.add_argument("--data", required=True, help="dataset directory (train/data.py layouts)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--step", type=int, default=None, help="default: the newest complete checkpoint")
    ap.add_argument("--device", default=None)
    ap.add_argument("--max_batches", type=int, default=None)
    ap.add_argument("--option", action="append", default=[], help="model option key=value (as in the job's file field)")
    a = ap.parse_args(argv)
    try:
        res = evaluate(a.model, a.export, a.data, a.batch, a.device, a.step, a.max_batches,
                       dict(o.split("=", 1) for o in a.option))
    except (FileNotFoundError, KeyError, ValueError) as e:
        print("evaluate: %s" % e, file=sys.stderr)
        return 1
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
