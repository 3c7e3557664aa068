"""TF1-shaped entry points for user training scripts.

The reference's jobs are TF1 between-graph scripts: the image's ``entry.sh`` runs the user's file with the task's nine
arguments (``/root/reference/handlers/train_handler.py:82-83``) and the script does what every distributed TF1 script
does -- ``ClusterSpec`` from the two host lists, a ``Server``, ``server.join()`` on the ps tasks,
``replica_device_setter`` + ``MonitoredTrainingSession(checkpoint_dir=export, is_chief=task_index == 0)`` on the workers
(SURVEY X2-X6).  Somebody who moves such a script here finds the same five names with the same roles; what is behind
them is this framework's runtime, not a translation of TF's:

=============================  ==========================================================================================
``ClusterSpec``                the job's ``{"ps": [...], "worker": [...]}`` host lists (argv / ``DTF_*_HOSTS``)
``Server``                     worker: joins the job's process group (RCCL over xGMI on GPUs, gloo on CPUs); ps: nothing
                               to serve in the colocated placement -- ``join()`` parks the task until the job is reaped
``replica_device_setter``      a ``Placement``: how many shard groups hold the variables (``ps_tasks``), not a device
                               function -- variables live in ONE flat buffer cut into shards (``parallel/layout.py``)
``SyncReplicasOptimizer``      marks an optimizer as synchronous (all workers' gradients reduced before the update);
                               without it an optimizer is applied with bounded staleness (``update_mode="stale"``),
                               the closest collective form of TF1's default asynchronous update
``MonitoredTrainingSession``   owns the ``ShardedPS`` runtime, restores the newest checkpoint of ``checkpoint_dir``,
                               saves every ``save_checkpoint_steps`` (all ranks write their shard, the chief the index),
                               runs hooks, and on exit drains the pipeline, writes the last checkpoint and leaves the group
``StopAtStepHook``             ``last_step`` / ``num_steps`` -> ``should_stop()``
=============================  ==========================================================================================

A training step is ``sess.run(loss)``: ``loss.backward()``, then the parameter-server step (push = reduce-scatter of the
gradient buckets, the owning shard applies the optimizer, pull = all-gather of the new variables).

``examples/mnist_replica.py`` is the classic TF1 sample written against this module; it runs by hand
(``--job_name=worker --task_index=0 --worker_hosts=...``) and as the ``file`` of a REST job.
"""
from __future__ import annotations

import os
import time

import torch

from ..parallel import checkpoint as ckpt
from ..parallel.comm import Comm
from ..parallel.ps_runtime import OptimizerSpec, ShardedPS


class ClusterSpec:
    """``ClusterSpec({"ps": ["h:2222"], "worker": ["h:2223", "h:2224"]})``; also from the two comma-separated lists a task
    gets as argv[3], argv[4], or from ``DTF_PS_HOSTS`` / ``DTF_WORKER_HOSTS``."""

    def __init__(self, cluster):
        if isinstance(cluster, ClusterSpec):
            cluster = cluster.as_dict()
        self._jobs = {str(j): [str(h) for h in (hosts.values() if isinstance(hosts, dict) else hosts)]
                      for j, hosts in dict(cluster).items()}

    @classmethod
    def from_lists(cls, ps_hosts="", worker_hosts=""):
        def split(s):
            return [h.strip() for h in (s or "").split(",") if h.strip()]

        return cls({"ps": split(ps_hosts), "worker": split(worker_hosts)})

    @classmethod
    def from_env(cls):
        return cls.from_lists(os.environ.get("DTF_PS_HOSTS", ""), os.environ.get("DTF_WORKER_HOSTS", ""))

    @classmethod
    def from_tf_config(cls, value=None):
        """-> ``(ClusterSpec, job_name, task_index)`` from a ``TF_CONFIG`` document (default: the environment's; the
        service sets it for every task): ``{"cluster": {"ps": [...], "worker": [...]}, "task": {"type", "index"}}``."""
        import json

        doc = json.loads(value if value is not None else os.environ.get("TF_CONFIG", "{}") or "{}")
        task = doc.get("task") or {}
        return cls(doc.get("cluster") or {}), str(task.get("type", "worker")), int(task.get("index", 0))

    @property
    def jobs(self):
        return sorted(self._jobs)

    def num_tasks(self, job_name):
        return len(self._jobs[job_name])

    def job_tasks(self, job_name):
        return list(self._jobs[job_name])

    def task_address(self, job_name, task_index):
        return self._jobs[job_name][int(task_index)]

    def as_dict(self):
        return {j: list(h) for j, h in self._jobs.items()}

    def __repr__(self):
        return "ClusterSpec(%r)" % self.as_dict()


class Server:
    """One task's membership in the job.  A worker joins the process group here (one rank per worker task); the
    rendezvous comes from the environment the service sets (``RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` /
    ``MASTER_PORT``) or, for a script started by hand, from the cluster: worker 0's address is the meeting point."""

    def __init__(self, cluster, job_name, task_index, device=None, backend=None, timeout_s=600):
        self.cluster = cluster if isinstance(cluster, ClusterSpec) else ClusterSpec(cluster)
        self.job_name, self.task_index = str(job_name), int(task_index)
        self.comm = None
        if self.job_name == "worker":
            n = max(1, self.cluster.num_tasks("worker")) if "worker" in self.cluster.jobs else 1
            if "RANK" not in os.environ and n > 1:
                host, _, port = self.cluster.task_address("worker", 0).rpartition(":")
                os.environ.setdefault("MASTER_ADDR", host or "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(int(port or 2222)))
                os.environ["RANK"], os.environ["WORLD_SIZE"] = str(self.task_index), str(n)
                os.environ.setdefault("LOCAL_RANK", str(self.task_index))
            self.comm = Comm.from_env(device=device, backend=backend, timeout_s=timeout_s)

    @property
    def target(self):
        c = self.comm
        return "%s://%s/%s:%d" % ((c.backend or "local") if c is not None else "parked",
                                  os.environ.get("MASTER_ADDR", "local"), self.job_name, self.task_index)

    @property
    def device(self):
        return self.comm.device if self.comm is not None else torch.device("cpu")

    def join(self):
        """ps task: block until the job is reaped (the cleaner deletes ps tasks once every worker is done --
        ``/root/reference/eventHandlers/CleanJobHandler.py:36-46``).  worker: wait for the other workers."""
        if self.job_name == "ps":
            from ..service.entry import ps_join_forever

            ps_join_forever()
        elif self.comm is not None:
            self.comm.barrier()


class Placement:
    """What ``replica_device_setter`` decides here: the number of shard groups that hold the variables."""

    def __init__(self, num_shards=None, ps_strategy="uniform"):
        self.num_shards, self.ps_strategy = num_shards, ps_strategy

    def __enter__(self):            # ``with tf.device(replica_device_setter(...)):`` has a body; so has this
        return self

    def __exit__(self, *exc):
        return False


def replica_device_setter(ps_tasks=0, cluster=None, worker_device=None, ps_strategy=None, **_):
    """TF1 assigns each variable to one ps task (round-robin or greedy by bytes).  Here every variable is a slice of one
    flat buffer that is cut into ``ps_tasks`` (``detail.ps``) equal shards, so the load is balanced by construction and
    ``ps_strategy`` has nothing left to decide; with ``ps_tasks`` >= the number of workers every worker holds one shard."""
    if not ps_tasks and cluster is not None:
        cs = cluster if isinstance(cluster, ClusterSpec) else ClusterSpec(cluster)
        ps_tasks = cs.num_tasks("ps") if "ps" in cs.jobs else 0
    return Placement(int(ps_tasks) or None)


class SyncReplicasOptimizer:
    """``SyncReplicasOptimizer(opt, replicas_to_aggregate=W)``: the update waits for every worker's gradient.  Partial
    aggregation (``replicas_to_aggregate`` < workers, TF1's backup workers) has no collective form and is refused."""

    def __init__(self, opt, replicas_to_aggregate=None, total_num_replicas=None):
        self.opt = _as_spec(opt)
        if replicas_to_aggregate is not None and total_num_replicas is not None \
                and int(replicas_to_aggregate) != int(total_num_replicas):
            raise ValueError("replicas_to_aggregate=%s of %s workers: backup workers are not supported (every worker's "
                             "gradient is part of the reduce-scatter)" % (replicas_to_aggregate, total_num_replicas))


def _as_spec(opt):
    if isinstance(opt, SyncReplicasOptimizer):
        return opt.opt
    if isinstance(opt, OptimizerSpec):
        return opt
    if isinstance(opt, dict):
        return OptimizerSpec(**opt)
    if isinstance(opt, str):
        return OptimizerSpec(kind=opt)
    if opt is None:
        return OptimizerSpec()
    raise TypeError("optimizer: an OptimizerSpec, a dict of its fields, a kind name or a SyncReplicasOptimizer, not %r" % (opt,))


class StopAtStepHook:
    def __init__(self, num_steps=None, last_step=None):
        if (num_steps is None) == (last_step is None):
            raise ValueError("exactly one of num_steps and last_step")
        self.num_steps, self.last_step = num_steps, last_step

    def begin(self, sess):
        if self.last_step is None:
            self.last_step = sess.global_step + int(self.num_steps)

    def after_run(self, sess, loss):
        if sess.global_step >= self.last_step:
            sess.request_stop()


class LoggingHook:
    """``every_n`` steps: one line with the step, the loss and the rate (this synchronises the device)."""

    def __init__(self, every_n=100, items_per_step=None, prefix="[train]"):
        self.every_n, self.items, self.prefix, self._t0, self._s0 = int(every_n), items_per_step, prefix, None, 0

    def begin(self, sess):
        self._t0, self._s0 = time.time(), sess.global_step

    def after_run(self, sess, loss):
        if sess.global_step % self.every_n == 0:
            v = float(loss.detach().float().item())
            dt = max(time.time() - self._t0, 1e-9)
            rate = (sess.global_step - self._s0) / dt
            print("%s step %d loss %.5f  %.1f steps/s%s" % (
                self.prefix, sess.global_step, v, rate,
                "  %.1f items/s" % (rate * self.items) if self.items else ""), flush=True)


class SummaryHook:
    """``tf.summary.scalar("loss", ...)`` + ``SummarySaverHook(save_steps=every_n)``: the chief appends the loss to
    ``<checkpoint_dir>/scalars.jsonl`` (``train/summary.py``) every ``every_n`` steps (this synchronises the device)."""

    def __init__(self, every_n=100, output_dir=None):
        self.every_n, self.output_dir, self._log = int(every_n), output_dir, None

    def begin(self, sess):
        from .summary import ScalarLog

        if sess.is_chief:
            self._log = ScalarLog(self.output_dir or sess.checkpoint_dir)

    def after_run(self, sess, loss):
        if self._log is not None and sess.global_step % self.every_n == 0:
            self._log.add(sess.global_step, loss=float(loss.detach().float().item()))


class MonitoredTrainingSession:
    """``with MonitoredTrainingSession(model, optimizer, server=server, checkpoint_dir=export, hooks=[...]) as sess:``
    ``while not sess.should_stop(): sess.run(loss_fn(model, next(batches)))``.

    * variables: ``model.named_parameters()`` are moved into the PS runtime's flat buffer (``ShardedPS``), sharded over
      ``placement.num_shards`` groups; worker 0's initial values are broadcast, as ``is_chief`` initialises in TF1;
    * restore: the newest complete checkpoint under ``checkpoint_dir`` (any worker count wrote it), ``global_step`` with it;
    * save: every ``save_checkpoint_steps`` steps and on exit; TF1-style variable names when ``tf_names`` maps them;
    * ``optimizer``: ``SyncReplicasOptimizer(...)`` -> synchronous update; a bare optimizer -> ``update_mode`` (default
      ``"sync"`` too: asynchronous-by-default is TF1's historical accident, ask for ``update_mode="stale"`` to get it)."""

    def __init__(self, model, optimizer=None, server=None, comm=None, is_chief=None, checkpoint_dir=None,
                 save_checkpoint_steps=None, hooks=(), placement=None, update_mode="sync", staleness=1, tf_names=None,
                 bucket_mb=32.0, wire_dtype=None, keep_checkpoints=5):
        self.model = model
        # the session leaves the group when it ends (a task that reaches interpreter exit with a live process group
        # aborts in the communicator's destructor); pass ``comm=`` to keep a communicator of your own alive
        self._own_comm, self._server = comm is None, server
        if comm is None and server is not None:
            comm = server.comm
        if comm is None:
            comm = Comm.from_env()
        self.comm = comm
        self.is_chief = (comm.rank == 0) if is_chief is None else bool(is_chief)
        if self.is_chief != (comm.rank == 0):
            raise ValueError("is_chief must be worker 0 (rank %d was given is_chief=%s): the chief writes the checkpoint "
                             "index and its variables are the initial ones" % (comm.rank, is_chief))
        if isinstance(optimizer, SyncReplicasOptimizer):
            update_mode = "sync"
        self.placement = placement or Placement()
        self.ps = ShardedPS(list(model.named_parameters()), comm, _as_spec(optimizer), bucket_mb=bucket_mb,
                            wire_dtype=wire_dtype, update_mode=update_mode, staleness=staleness,
                            num_shards=self.placement.num_shards)
        if hasattr(self.ps, "attach"):
            self.ps.attach(model)
        self.checkpoint_dir = checkpoint_dir or None
        self.save_checkpoint_steps = int(save_checkpoint_steps) if save_checkpoint_steps else 0
        self.tf_names = tf_names or {}
        self.hooks = list(hooks)
        self._stop = False
        self._last_saved = None
        if keep_checkpoints is not None:
            ckpt.KEEP = int(keep_checkpoints)
        if self.checkpoint_dir:
            os.makedirs(self.checkpoint_dir, exist_ok=True)
            if ckpt.latest_step(self.checkpoint_dir) is not None:
                from .trainer import restore_checkpoint

                step = restore_checkpoint(model, self.ps, self.checkpoint_dir, self.tf_names)
                self.ps.global_step = int(step)
                self._last_saved = int(step)
                if self.is_chief:
                    print("[session] restored %s, global_step %d" % (self.checkpoint_dir, step), flush=True)

    # -- the TF1 surface ------------------------------------------------------------------------------------------------
    @property
    def global_step(self):
        return int(self.ps.global_step)

    def should_stop(self):
        return self._stop

    def request_stop(self):
        self._stop = True

    def run(self, loss):
        """One training step for the loss of this worker's batch; returns the loss tensor (reading its value
        synchronises the device: do that every N steps, not every step)."""
        if not self._begun:
            self._begin()
        loss.backward()
        self.ps.step()
        if self.save_checkpoint_steps and self.checkpoint_dir and self.global_step % self.save_checkpoint_steps == 0:
            self.save()
        for h in self.hooks:
            if hasattr(h, "after_run"):
                h.after_run(self, loss)
        return loss

    _begun = False

    def _begin(self):
        self._begun = True
        for h in self.hooks:
            if hasattr(h, "begin"):
                h.begin(self)

    def save(self):
        """Every rank writes its shard, the chief the replicated state and the index (call it on all ranks)."""
        if not self.checkpoint_dir or self._last_saved == self.global_step:
            return
        from .trainer import save_checkpoint

        if hasattr(self.ps, "flush"):
            self.ps.flush()
        save_checkpoint(self.model, self.ps, self.checkpoint_dir, self.global_step, self.tf_names, self.is_chief)
        self._last_saved = self.global_step

    def __enter__(self):
        if not self._begun:
            self._begin()
        return self

    def __exit__(self, et, ev, tb):
        try:
            if et is None:
                if hasattr(self.ps, "flush"):
                    self.ps.flush()
                self.save()
                for h in self.hooks:
                    if hasattr(h, "end"):
                        h.end(self)
                self.comm.barrier()
        finally:
            if self._own_comm:
                # an exception on this rank: peers may sit in a collective this rank never joins -- bounded destroy
                self.comm.close(timeout_s=None if et is None else 15.0)
                if self._server is not None:
                    self._server.comm = None
        return False
