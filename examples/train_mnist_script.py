"""A user training script for a job whose ``file`` field names a python file -- the reference's mode of use: its image
runs the user's TF1 script with the task's nine arguments (/root/reference/handlers/train_handler.py:82-83:
job_name, task_index, ps_hosts, worker_hosts, file, data, export, hdfs web, hdfs namenode).  Here the same arguments arrive
in ``sys.argv[1:]`` and the worker rendezvous in the environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), so a script
only has to join and train:

    curl -d '{"file": "/path/to/examples/train_mnist_script.py", "data": "/data/mnist", "export": "/tmp/out",
              "detail": {"ps": 1, "worker": 2}}' http://127.0.0.1:12345/v1/train

The between-graph MNIST MLP of the reference's sample job on the framework's public pieces: ``Comm`` (RCCL on GPUs, gloo on
CPUs), ``ShardedPS`` (variables sharded over the ranks, push = reduce-scatter, pull = all-gather), ``train.data`` (the
``data`` directory: MNIST IDX files, sharded over the workers and prefetched; synthetic digits if it holds none)."""
import json
import os
import sys

import torch

from distributed_tensorflow_amd.models import mnist_mlp
from distributed_tensorflow_amd.parallel.comm import Comm
from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS
from distributed_tensorflow_amd.train import data as D


def main(argv):
    job_name, task_index = argv[0], int(argv[1])
    data_dir, export = argv[5], argv[6]
    steps, batch = int(os.environ.get("STEPS", "60")), int(os.environ.get("BATCH", "64"))
    comm = Comm.from_env()                    # one rank per worker task; ps tasks never get here (the service parks them)
    dev = comm.device
    torch.manual_seed(0)                      # same initial variables on every worker; the PS broadcasts rank 0's anyway
    model = mnist_mlp.build().to(dev)
    ps = ShardedPS(list(model.named_parameters()), comm, OptimizerSpec(kind="momentum", lr=0.05, momentum=0.9))
    ds = D.open_dataset(data_dir)
    if ds is not None:
        batches = D.Prefetcher(D.BatchStream(ds, batch, worker=comm.rank, num_workers=comm.world_size, seed=0),
                               D.transform_for(mnist_mlp, ds, dev), dev)
    else:
        batches = D.Cycle([mnist_mlp.synthetic_batch(batch, device=dev, seed=100 * comm.rank + k) for k in range(8)])
    losses = []
    for _ in range(steps):
        loss = mnist_mlp.loss_fn(model, next(batches))
        loss.backward()
        ps.step()                             # push gradients, apply on the owning shard, pull the new variables
        losses.append(float(loss))
    batches.close()
    print("[%s %d] %d steps on %s, loss %.4f -> %.4f (%s)" % (job_name, task_index, steps, dev, losses[0], losses[-1],
                                                             "dataset %s" % ds.source if ds is not None else "synthetic"), flush=True)
    if comm.rank == 0 and export:
        os.makedirs(export, exist_ok=True)
        with open(os.path.join(export, "result.json"), "w") as f:
            json.dump({"first_loss": losses[0], "final_loss": losses[-1], "steps": steps, "workers": comm.world_size,
                       "dataset": None if ds is None else len(ds)}, f)
    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main(sys.argv[1:])
