"""The classic TF1 distributed sample (between-graph replication of an MNIST MLP: ClusterSpec, Server, ps tasks join,
workers build under replica_device_setter and train in a MonitoredTrainingSession) written against
``distributed_tensorflow_amd.train.session`` -- the shape of script the reference's image runs for a job
(/root/reference/handlers/train_handler.py:82-83 passes it job_name, task_index, ps_hosts, worker_hosts, file, data, export,
hdfs web, hdfs namenode).

By hand, two workers on one box (ps tasks have nothing to serve in the colocated placement and may be left out):

    python examples/mnist_replica.py --job_name=worker --task_index=0 --worker_hosts=127.0.0.1:2223,127.0.0.1:2224 &
    python examples/mnist_replica.py --job_name=worker --task_index=1 --worker_hosts=127.0.0.1:2223,127.0.0.1:2224

As a job of the service (the nine positional arguments arrive in sys.argv):

    curl -d '{"file": "/path/to/examples/mnist_replica.py", "data": "/data/mnist", "export": "/tmp/out",
              "detail": {"ps": 1, "worker": 2}}' http://127.0.0.1:12345/v1/train
"""
import argparse
import os
import sys

import torch

from distributed_tensorflow_amd.models import mnist_mlp
from distributed_tensorflow_amd.train import data as D
from distributed_tensorflow_amd.train.session import (ClusterSpec, LoggingHook, MonitoredTrainingSession, Server,
                                                      StopAtStepHook, SyncReplicasOptimizer, replica_device_setter)


def parse(argv):
    p = argparse.ArgumentParser()
    p.add_argument("--job_name", default="worker")
    p.add_argument("--task_index", type=int, default=0)
    p.add_argument("--ps_hosts", default="")
    p.add_argument("--worker_hosts", default="")
    p.add_argument("--data_dir", default="")
    p.add_argument("--train_dir", default="")
    p.add_argument("--train_steps", type=int, default=int(os.environ.get("STEPS", "200")))
    p.add_argument("--batch_size", type=int, default=int(os.environ.get("BATCH", "100")))
    p.add_argument("--hidden_units", type=int, default=100)
    p.add_argument("--learning_rate", type=float, default=0.05)
    p.add_argument("--sync_replicas", type=int, default=1)
    if argv and not argv[0].startswith("-"):
        # the service's positional form: job_name task_index ps_hosts worker_hosts file data export [hdfs web, namenode]
        a = p.parse_args([])
        a.job_name, a.task_index, a.ps_hosts, a.worker_hosts = argv[0], int(argv[1]), argv[2], argv[3]
        a.data_dir, a.train_dir = argv[5], argv[6]
        return a
    return p.parse_args(argv)


def main(argv):
    F = parse(argv)
    cluster = ClusterSpec.from_lists(F.ps_hosts, F.worker_hosts)
    server = Server(cluster, job_name=F.job_name, task_index=F.task_index)
    if F.job_name == "ps":
        server.join()
        return 0
    dev = server.device
    num_workers = max(1, cluster.num_tasks("worker"))
    torch.manual_seed(0)
    with replica_device_setter(cluster=cluster) as placement:
        model = mnist_mlp.build(hidden_units=F.hidden_units).to(dev)
    opt = {"kind": "momentum", "lr": F.learning_rate, "momentum": 0.9}
    if F.sync_replicas:
        opt = SyncReplicasOptimizer(opt, replicas_to_aggregate=num_workers, total_num_replicas=num_workers)
    ds = D.open_dataset(F.data_dir) if F.data_dir else None
    hooks = [StopAtStepHook(last_step=F.train_steps), LoggingHook(every_n=50, items_per_step=F.batch_size * num_workers)]
    with MonitoredTrainingSession(model, opt, server=server, is_chief=(F.task_index == 0),
                                  checkpoint_dir=F.train_dir or None, save_checkpoint_steps=100, hooks=hooks,
                                  placement=placement, update_mode="sync" if F.sync_replicas else "stale",
                                  tf_names=mnist_mlp.tf_variable_names(model)) as sess:
        if ds is not None:
            batches = D.Prefetcher(D.BatchStream(ds, F.batch_size, worker=F.task_index, num_workers=num_workers, seed=0,
                                                 start_step=sess.global_step),
                                   D.transform_for(mnist_mlp, ds, dev), dev)
        else:
            batches = D.Cycle([mnist_mlp.synthetic_batch(F.batch_size, device=dev, seed=100 * F.task_index + k)
                               for k in range(8)])
        first = last = None
        try:
            while not sess.should_stop():
                last = sess.run(mnist_mlp.loss_fn(model, next(batches)))
                if first is None:
                    first = float(last)
        finally:
            batches.close()
        print("[worker %d] global_step %d on %s, loss %s -> %s (%s)" % (
            F.task_index, sess.global_step, dev, "%.4f" % first if first is not None else "-",
            "%.4f" % float(last) if last is not None else "-", ds.source if ds is not None else "synthetic"), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
