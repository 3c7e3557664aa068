"""Communicator layer (new layer N1; replaces external X2, the TF1 gRPC
``tf.train.Server`` mesh on port 2222 that the reference's headless Services
address, ``handlers/train_handler.py:31-35,84,99-100``).

One process per GPU; ``torch.distributed`` with backend ``"nccl"`` **is RCCL** on
ROCm and rides xGMI inside the node; ``"gloo"`` is the CPU plumbing backend of
config #1 and of the multi-process tests.  The PS traffic maps onto collectives:

* push  = ``reduce_scatter``  (worker -> owning PS shard), or the *direct* form
  ``all_to_all`` + W-way sum fused into the shard-apply kernel: on MI355X's
  point-to-point xGMI mesh (7 links x ~153 GB/s per GPU) the all-to-all puts
  exactly one slice on every link at once, whereas a ring reduce-scatter is
  bound by a single link;
* pull  = ``all_gather``      (PS shard -> every worker);
* embedding id/row exchange = ``all_to_all_single`` with splits;
* global_step / grad-norm / metrics = scalar ``all_reduce``.

``Comm`` with ``world_size == 1`` needs no process group at all (1-GPU bench).
"""
from __future__ import annotations

import datetime
import os

import torch
import torch.distributed as dist


class Comm:
    def __init__(self, rank=0, world_size=1, device=None, group=None, backend=None, live=None):
        self.rank, self.world_size = rank, world_size
        # live: collectives really go to the backend.  A 1-rank communicator is normally a
        # no-op; ``from_env(force=True)`` makes it live so that the whole RCCL call path
        # (streams, in-place slices, bf16) can be exercised on a single GPU.
        self.live = (world_size > 1) if live is None else bool(live)
        self.device = device or torch.device("cpu")
        self.group = group
        self.backend = backend or ("none" if not self.live else dist.get_backend(group))

    # -- construction ---------------------------------------------------------
    @classmethod
    def single(cls, device=None):
        return cls(0, 1, device)

    @classmethod
    def from_env(cls, device=None, backend=None, timeout_s=600, force=False):
        """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
        (set by torchrun, or by ``service/entry.py`` from the job's host lists)."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        if device is None:
            if torch.cuda.is_available():
                device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
            else:
                device = torch.device("cpu")
        if device.type == "cuda":
            torch.cuda.set_device(device)
        if world == 1 and not force:
            return cls(0, 1, device)
        if backend == "rccl-native":
            # direct RCCL calls on the caller's stream (native/rccl_comm.cpp): no c10d Work objects, no watchdog
            from .rccl_native import NativeComm

            return NativeComm.from_env(device, timeout_s)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if not dist.is_initialized():
            backend = backend or ("nccl" if device.type == "cuda" else "gloo")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = device
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=timeout_s), **kw)
        return cls(dist.get_rank(), dist.get_world_size(), device, None, dist.get_backend(), live=True)

    def subgroup(self, ranks):
        """Communicator over a subset of ranks (every rank must call this)."""
        g = dist.new_group(ranks)
        if self.rank not in ranks:
            return None
        return Comm(ranks.index(self.rank), len(ranks), self.device, g, self.backend, live=True)

    # -- collectives ------------------------------------------------------------
    def reduce_scatter(self, out_shard, in_full, async_op=False):
        if not self.live:
            if out_shard.data_ptr() != in_full.data_ptr():
                out_shard.copy_(in_full[:out_shard.numel()])
            return None
        return dist.reduce_scatter_tensor(out_shard, in_full, group=self.group, async_op=async_op)

    def all_gather(self, out_full, in_shard, async_op=False):
        if not self.live:
            if out_full.data_ptr() != in_shard.data_ptr():
                out_full[:in_shard.numel()].copy_(in_shard)
            return None
        return dist.all_gather_into_tensor(out_full, in_shard, group=self.group, async_op=async_op)

    def all_to_all(self, out, inp, out_splits=None, in_splits=None, async_op=False):
        if not self.live:
            out.copy_(inp)
            return None
        return dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=async_op)

    def all_reduce(self, t, op=dist.ReduceOp.SUM, async_op=False):
        if not self.live:
            return None
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def broadcast(self, t, src=0, async_op=False):
        if not self.live:
            return None
        return dist.broadcast(t, src=self._global(src), group=self.group, async_op=async_op)

    def reduce(self, t, dst=0, async_op=False):
        if not self.live:
            return None
        return dist.reduce(t, dst=self._global(dst), group=self.group, async_op=async_op)

    def send(self, t, dst, tag=0):
        return dist.send(t, self._global(dst), group=self.group, tag=tag)

    def isend(self, t, dst, tag=0):
        return dist.isend(t, self._global(dst), group=self.group, tag=tag)

    def irecv(self, t, src, tag=0):
        """Posted receive from ONE peer (works on RCCL and gloo; ``recv(src=None)`` is gloo-only)."""
        return dist.irecv(t, self._global(src), group=self.group, tag=tag)

    def recv(self, t, src=None, tag=0):
        r = dist.recv(t, None if src is None else self._global(src), group=self.group, tag=tag)
        return self._local(r)

    def barrier(self):
        if self.live:
            if self.device.type == "cuda" and "nccl" in str(self.backend):
                dist.barrier(group=self.group, device_ids=[self.device.index])
            else:
                dist.barrier(group=self.group)

    def _global(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _local(self, r):
        return r if self.group is None else dist.get_group_rank(self.group, r)

    def close(self, timeout_s=None, on_timeout_exit=None):
        """Tear the process group down (``destroy_process_group``).  A task that reaches interpreter exit with a
        live gloo / RCCL group dies in ``std::terminate`` ("terminate called without an active exception": the
        backend's worker threads are still joinable when the static destructors run) -- a rank that had FINISHED
        its job then reports a failure exit code, and the reference's lifecycle keys everything off
        ``status.succeeded == 1`` (``/root/reference/eventHandlers/CleanJobHandler.py:48-59``).

        ``timeout_s`` + ``on_timeout_exit``: on an error path the peers may be gone and the destroy may block
        (RCCL waits for outstanding work); after ``timeout_s`` the process ends with ``on_timeout_exit``."""
        if not (self.live and dist.is_initialized()):
            self.live = False
            return
        timer = None
        if timeout_s and on_timeout_exit is not None:
            import sys
            import threading

            def _bail():
                try:
                    sys.stdout.flush()
                    sys.stderr.flush()
                finally:
                    os._exit(int(on_timeout_exit))

            timer = threading.Timer(float(timeout_s), _bail)
            timer.daemon = True
            timer.start()
        try:
            if self.group is None:
                dist.destroy_process_group()
            else:
                dist.destroy_process_group(self.group)
        finally:
            if timer is not None:
                timer.cancel()
            self.live = False
