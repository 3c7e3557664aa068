"""Dedicated parameter-server placement: PS tasks are their own processes
(TF1 between-graph replication as the reference's job spec describes it:
``detail.ps`` PS tasks + ``detail.worker`` workers, ``post-data:10-13``).

Rank order inside the job's process group: ``ps-0 .. ps-(P-1), worker-0 ..
worker-(W-1)``.  The flat variable space is cut into P shards
(:class:`~.layout.FlatLayout` with ``world = P``); PS task j hosts shard j
(fp32 master + optimizer slots) and runs :meth:`PSServer.serve` forever --
like ``tf.train.Server.join()`` it never exits on its own, which is exactly why
the reference needs ``remove_ps.py`` to reap it.

sync   (``SyncReplicasOptimizer`` semantics): per step one control all-reduce
       (done flag), then per (bucket, PS j) ``reduce(dst=j)`` of the workers'
       gradient slices (PS ranks add zeros) -> shard apply on the PS ->
       ``broadcast(src=j)`` of the updated slice.
async  (TF1 default): each worker talks to every PS point-to-point on its own
       clock: ``send(grad slice + [flag, step])`` then ``recv(updated slice)``;
       the PS applies each worker's gradient the moment it arrives (no
       averaging, no barrier).  The PS keeps ONE posted ``irecv`` per worker and
       polls the requests, serving workers in arrival order: no ``recv(src=ANY)``
       (which only gloo has), so the same loop runs over RCCL when the PS tasks
       own GPUs (``resource.ps_gpu``) -- there every peer pair has its own
       communicator stream, and a worker that has not sent yet blocks nobody.
When every worker has signalled done the PS leaves its loop, optionally writes
its checkpoint shard, and blocks until it is killed.

Checkpoints: PS task j writes shard j (master + slots), the chief writes the
replicated state and -- after a barrier -- the index that publishes the checkpoint.  At the
end of training in both modes; in sync mode also every ``ckpt_every`` steps (the PS and the
workers are in lockstep there, so "after step s" is the same moment everywhere).  Resume
(``restore``): the PS tasks load master + slots of THEIR shard from a checkpoint written by
any placement / task count, the workers load the variables they compute with.
"""
from __future__ import annotations

import torch

from ..ops import bucket as bucket_ops
from ..ops import optim as optim_ops
from .comm import Comm
from .layout import FlatLayout
from .ps_runtime import OptimizerSpec, trainable

PUSH, DONE = 0.0, 1.0


def _apply(opt, g, master, m, v, p_out, lr, step, grad_scale=1.0):
    if opt.kind == "adam":
        optim_ops.adam_apply(g, master, m, v, p_out, lr=lr, beta1=opt.beta1, beta2=opt.beta2, eps=opt.eps,
                             wd=opt.weight_decay, adamw=opt.adamw, bias_correction=opt.bias_correction,
                             step=step, grad_scale=grad_scale)
    else:
        mom = opt.momentum if opt.kind == "momentum" else 0.0
        optim_ops.sgd_momentum_apply(g, master, m, p_out, lr=lr, momentum=mom, wd=opt.weight_decay,
                                     nesterov=opt.nesterov, grad_scale=grad_scale)


class _Base:
    def __init__(self, named_shapes, comm: Comm, num_ps, opt, bucket_mb, wire_dtype, device):
        self.comm, self.P = comm, int(num_ps)
        self.W = comm.world_size - self.P
        assert self.P >= 1 and self.W >= 1
        self.opt = opt or OptimizerSpec()
        self.wire_dtype = wire_dtype
        self.device = device
        esize = torch.empty((), dtype=wire_dtype).element_size()
        self.layout = FlatLayout(named_shapes, self.P, max(1, int(bucket_mb * (1 << 20) / esize)))


class PSServer(_Base):
    """Runs inside a ``ps`` task."""

    def __init__(self, named_shapes, comm, num_ps, opt=None, bucket_mb=32.0, wire_dtype=torch.float32,
                 device=None, update_mode="sync"):
        super().__init__(named_shapes, comm, num_ps, opt, bucket_mb, wire_dtype, device or torch.device("cpu"))
        assert comm.rank < self.P, "PSServer must run on a ps rank"
        self.j = comm.rank
        self.update_mode = update_mode
        L, dev = self.layout, self.device
        self.master = torch.zeros(L.shard_total, dtype=torch.float32, device=dev)
        self.slot_m = torch.zeros(L.shard_total, dtype=torch.float32, device=dev)
        self.slot_v = torch.zeros(L.shard_total, dtype=torch.float32, device=dev) if self.opt.kind == "adam" else None
        self.wire = torch.zeros(L.shard_total, dtype=wire_dtype, device=dev)
        self.global_step = 0
        self.lr = self.opt.lr

    def _init_from_chief(self, adopt=True):
        """Initial values come from worker 0 (the chief), like TF's chief
        running the init op against the PS-hosted variables.  ``adopt=False`` (resumed job): the broadcasts still
        run -- the workers are in them -- but the restored master is kept."""
        L = self.layout
        chief = self.P
        for bk in L.buckets:
            for j in range(self.P):
                n = bk.shard
                buf = torch.zeros(n, dtype=self.wire_dtype, device=self.device)
                self.comm.broadcast(buf, chief)
                if j == self.j and adopt:
                    self.master[bk.shard_off:bk.shard_off + n].copy_(buf)

    # what checkpoint.restore_ps needs of a "ps" object: this task hosts shard j of P and computes with nothing
    param_flat = None
    hosted = True

    def restore(self, export_dir, named_meta_params, tf_names):
        """Resume: load master + optimizer slots of shard j (re-partitioned from whatever wrote the checkpoint)."""
        from . import checkpoint as ckpt

        self.params = [p for _, p in named_meta_params]
        rd = ckpt.restore_ps(self, export_dir, tf_names=tf_names)
        self.global_step = rd.step
        return rd

    def serve(self, ckpt_every=0, write_shard=None, skip_init=False):
        """``ckpt_every`` / ``write_shard(step)``: sync mode writes shard j every that many steps and then meets the
        workers in a barrier (the chief publishes the index after it)."""
        self._init_from_chief(adopt=not skip_init)
        if self.update_mode == "async":
            self._serve_async()
        else:
            self._serve_sync(int(ckpt_every or 0), write_shard)
        return self.global_step

    def _serve_sync(self, ckpt_every=0, write_shard=None):
        L = self.layout
        ctrl = torch.zeros(2, dtype=torch.float32, device=self.device)
        first_step = self.global_step
        while True:
            ctrl.zero_()
            self.comm.all_reduce(ctrl)
            if ctrl[0].item() > 0:
                return
            for bk in L.buckets:
                for j in range(self.P):
                    g = torch.zeros(bk.shard, dtype=self.wire_dtype, device=self.device)
                    self.comm.reduce(g, j)
                    sl = slice(bk.shard_off, bk.shard_off + bk.shard)
                    if j == self.j:
                        _apply(self.opt, g, self.master[sl], self.slot_m[sl],
                               None if self.slot_v is None else self.slot_v[sl], self.wire[sl], self.lr,
                               self.global_step + 1)
                        self.comm.broadcast(self.wire[sl], j)
                    else:
                        self.comm.broadcast(g, j)
            self.global_step += 1
            if ckpt_every and write_shard is not None and self.global_step % ckpt_every == 0 \
                    and self.global_step > first_step:
                write_shard(self.global_step)
                self.comm.barrier()         # the workers wait here too; the chief then writes the index

    def _serve_async(self, poll_s=50e-6):
        """One posted ``irecv`` per worker, served in arrival order.

        RCCL requests are polled (``is_completed`` is an event query there).  A
        gloo request only completes inside ``wait()``, so on gloo one waiter
        thread per worker blocks in ``wait()`` and reports the arrival through a
        queue; the apply loop itself stays single-threaded either way."""
        import queue
        import threading
        import time

        n = self.layout.shard_total
        W, P = self.W, self.P
        inbox = [torch.zeros(n + 2, dtype=self.wire_dtype, device=self.device) for _ in range(W)]
        outbox = [torch.zeros(n, dtype=self.wire_dtype, device=self.device) for _ in range(W)]
        recv = [self.comm.irecv(inbox[w], P + w, tag=2) for w in range(W)]
        sent = [None] * W
        polled = self.comm.backend == "nccl"
        arrived = queue.Queue()

        def waiter(w, req):
            try:
                req.wait()
                arrived.put((w, None))
            except Exception as e:                  # surfaced in the serving thread
                arrived.put((w, e))

        def post(w):
            recv[w] = self.comm.irecv(inbox[w], P + w, tag=2)
            if not polled:
                threading.Thread(target=waiter, args=(w, recv[w]), daemon=True).start()

        def ready():
            if not polled:
                w, err = arrived.get()
                if err is not None:
                    raise err
                return [w]
            out = [w for w in range(W) if recv[w] is not None and recv[w].is_completed()]
            for w in out:
                recv[w].wait()                      # orders the device work after the receive
            if not out:
                time.sleep(poll_s)
            return out

        if not polled:
            for w in range(W):
                threading.Thread(target=waiter, args=(w, recv[w]), daemon=True).start()
        live = W
        while live:
            for w in ready():
                if float(inbox[w][n].item()) == DONE:
                    recv[w] = None
                    live -= 1
                    continue
                _apply(self.opt, inbox[w][:n], self.master, self.slot_m, self.slot_v, self.wire, self.lr,
                       self.global_step + 1)
                self.global_step += 1
                if sent[w] is not None:
                    sent[w].wait()                  # the previous reply to this worker has left its buffer
                outbox[w].copy_(self.wire)
                sent[w] = self.comm.isend(outbox[w], P + w, tag=3)
                post(w)
        for h in sent:
            if h is not None:
                h.wait()


def agree_on_step(comm, mine):
    """Everyone learns the number of applies the PS tasks performed (workers pass 0)."""
    t = torch.tensor([float(mine)], dtype=torch.float32, device=comm.device if comm.backend == "nccl" else "cpu")
    comm.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return int(t.item())


class PSWorkerClient(_Base):
    """Runs inside a ``worker`` task; same ``step()`` surface as ShardedPS."""

    def __init__(self, named_params, comm, num_ps, opt=None, bucket_mb=32.0, wire_dtype=torch.float32,
                 update_mode="sync"):
        self.named_params = trainable(named_params)
        self.params = [p for _, p in self.named_params]
        dev = self.params[0].device
        super().__init__([(n, p.numel()) for n, p in self.named_params], comm, num_ps, opt, bucket_mb,
                         wire_dtype, dev)
        assert comm.rank >= self.P, "PSWorkerClient must run on a worker rank"
        self.w = comm.rank - self.P
        self.update_mode = update_mode
        L = self.layout
        self.param_flat = torch.zeros(L.total, dtype=wire_dtype, device=dev)
        self.grad_flat = torch.zeros(L.total, dtype=wire_dtype, device=dev)
        self._plans = [bucket_ops.PackPlan([L.slots[i].numel for i in bk.params],
                                           [L.slots[i].flat_off for i in bk.params]) for bk in L.buckets]
        self.zero_copy = []
        for slot, p in zip(L.slots, self.params):
            self.param_flat[slot.flat_off:slot.flat_off + slot.numel].copy_(bucket_ops.dense_storage_view(p.data))
        # chief initialises the PS-hosted variables; everyone adopts them
        chief = self.P
        for bk in L.buckets:
            for j in range(self.P):
                lo, hi = L.shard_range(bk.index, j)
                self.comm.broadcast(self.param_flat[lo:hi], chief)
        for slot, p in zip(L.slots, self.params):
            zc = p.dtype == wire_dtype
            if zc:
                p.data = torch.as_strided(self.param_flat, p.shape, p.stride(), slot.flat_off)
            self.zero_copy.append(zc)
        self.global_step = 0
        self._sync_params()

    def _sync_params(self):
        for b, bk in enumerate(self.layout.buckets):
            idx = [i for i in bk.params if not self.zero_copy[i]]
            if idx:
                plan = bucket_ops.PackPlan([self.layout.slots[i].numel for i in idx],
                                           [self.layout.slots[i].flat_off for i in idx])
                plan.unpack(self.param_flat, [self.params[i].data for i in idx])

    def _pack_all(self, scale):
        for b, bk in enumerate(self.layout.buckets):
            grads = []
            for i in bk.params:
                p = self.params[i]
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                grads.append(p.grad if p.grad.stride() == p.stride() else torch.empty_like(p).copy_(p.grad))
            self._plans[b].pack(grads, self.grad_flat, scale)

    def step(self):
        L = self.layout
        if self.update_mode == "async":
            self._pack_all(1.0)
            tail = torch.tensor([PUSH, 0.0], dtype=self.wire_dtype, device=self.device)
            for j in range(self.P):
                self.comm.send(torch.cat([self._shard_view(self.grad_flat, j), tail]), j, tag=2)
            for j in range(self.P):
                buf = torch.empty(L.shard_total, dtype=self.wire_dtype, device=self.device)
                self.comm.recv(buf, j, tag=3)
                self._scatter_shard(buf, j)
        else:
            self._pack_all(1.0 / self.W)
            ctrl = torch.tensor([0.0, 1.0], dtype=torch.float32, device=self.device)
            self.comm.all_reduce(ctrl)
            for bk in L.buckets:
                for j in range(self.P):
                    lo, hi = L.shard_range(bk.index, j)
                    self.comm.reduce(self.grad_flat[lo:hi], j)
                    self.comm.broadcast(self.param_flat[lo:hi], j)
        self._sync_params()
        for p in self.params:
            p.grad = None
        self.global_step += 1

    def _shard_view(self, flat, j):
        """Shard j as one contiguous tensor (concatenation of its bucket slices)."""
        L = self.layout
        if len(L.buckets) == 1:
            lo, hi = L.shard_range(0, j)
            return flat[lo:hi]
        return torch.cat([flat[slice(*L.shard_range(bk.index, j))] for bk in L.buckets])

    def _scatter_shard(self, buf, j):
        L = self.layout
        for bk in L.buckets:
            lo, hi = L.shard_range(bk.index, j)
            self.param_flat[lo:hi].copy_(buf[bk.shard_off:bk.shard_off + bk.shard])

    def finish(self):
        """Tell the PS tasks this worker is done (the 'signal from worker' the
        reference's comment at ``eventHandlers/CleanJobHandler.py:42`` alludes to)."""
        if self.update_mode == "async":
            msg = torch.zeros(self.layout.shard_total + 2, dtype=self.wire_dtype, device=self.device)
            msg[-2] = DONE
            for j in range(self.P):
                self.comm.send(msg, j, tag=2)
        else:
            ctrl = torch.tensor([1.0, 1.0], dtype=torch.float32, device=self.device)
            self.comm.all_reduce(ctrl)
        return agree_on_step(self.comm, 0)

    def flush(self):
        pass

    # what checkpoint.restore_ps needs: a worker hosts no PS state, it only refreshes what it computes with
    hosted = False
    master = slot_m = slot_v = torch.zeros(0)

    def restore(self, export_dir, tf_names):
        from . import checkpoint as ckpt

        rd = ckpt.restore_ps(self, export_dir, tf_names=tf_names)
        self._sync_params()
        return rd

    def _unpack_params(self):
        self._sync_params()

    def set_lr(self, lr):
        # the learning rate lives on the PS in this placement
        pass
