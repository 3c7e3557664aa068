"""True asynchronous parameter-server update for colocated ranks (SURVEY 2.3 "Async PS update",
7.2 item 3c; the TF1 default the reference's jobs run -- its PS tasks never synchronise the
workers, ``eventHandlers/CleanJobHandler.py:41-43``, ``handlers/train_handler.py:96-112``).

Collectives are bulk-synchronous, TF1's asynchronous PS is not, so this mode uses NO collective
in the training step.  Every rank hosts shard r of the flat variable space (fp32 master +
optimizer slots + the shard's hyper-parameters) in memory that all the other ranks map:

* GPU: ``hipMalloc`` + ``hipIpcGetMemHandle`` on the owner, ``hipIpcOpenMemHandle`` on the
  peers (dmabuf IPC; the mapping is reached over xGMI when the owner is another GPU);
* CPU (tests, config #1 plumbing): one mmap-ed file per shard.

A worker's step is then ONE-SIDED: for every owner j it launches the K1 / K2 fused shard-apply
kernel on *its own* device with ``g`` = its local gradient slice j, ``p32 / m / v`` = owner j's
state (peer memory, read-modify-write over the fabric) and ``p_out`` = its local wire-dtype
parameter slice j -- push, apply and pull of slice j in a single kernel, with no rendezvous
with any other worker.  Workers run on their own clocks; concurrent applies to one shard race
exactly like TF1's ``use_locking=False`` updates (Hogwild), each worker's gradient is applied
un-averaged with the full learning rate, and each owner counts the applies it received
(TF's ``global_step``).  Visibility is kernel-boundary granular: a worker's next forward reads
the values its own apply kernels wrote, other workers' updates are picked up at its next apply.

Owners are visited in rotated order (rank r starts with shard r, then r+1, ...), so at any
moment the W workers tend to target W different owners: every xGMI link carries one stream.

The packs and applies are hook-driven like the synchronous runtime's: when the last gradient
of a bucket has been accumulated, the bucket is packed and applied to every owner on a second
stream while backward continues with the earlier layers (``DTF_ASYNC_HOOKS=0``: everything
after backward, the first version).  With global-norm clipping the applies need the norm of
the whole gradient, so only the packs move into backward.

The torch.distributed group is used for control only (handle exchange, final barrier): gloo is
enough, which also makes the mode testable with several processes on ONE GPU.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import tempfile

import torch
import torch.distributed as dist

from ..ops import bucket as bucket_ops
from ..ops import hip
from ..ops import optim as optim_ops
from ..ops import reduce as reduce_ops
from ..ops import streams as streams_mod
from .comm import Comm
from .layout import FlatLayout
from .ps_runtime import OptimizerSpec, trainable

_HYPER = 8      # floats reserved at the head of a shard region: [lr, grad scale, applies+1, ...]


class _DevicePtr:
    """``__cuda_array_interface__`` shim: a torch view of device memory torch did not allocate."""

    def __init__(self, ptr, numel):
        self.__cuda_array_interface__ = {"shape": (int(numel),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2}


def _declare_ipc():
    P, L = ctypes.c_void_p, ctypes.c_int64
    hip.declare("dtf_ipc_handle_bytes", [])
    hip.declare("dtf_ipc_alloc", [ctypes.POINTER(P), L])
    hip.declare("dtf_ipc_free", [P])
    hip.declare("dtf_ipc_get_handle", [P, P])
    hip.declare("dtf_ipc_open", [P, ctypes.POINTER(P)])
    hip.declare("dtf_ipc_close", [P])
    return hip.lib()


class _SharedRegions:
    """Region r = ``nfloat`` fp32 values owned by rank r and mapped by everyone."""

    def __init__(self, comm, device, nfloat, shm_dir=None):
        self.comm, self.device, self.n = comm, device, int(nfloat)
        self.views = [None] * comm.world_size
        self._own_ptr = None
        self._peer_ptrs = []
        self._files = []
        W, r = comm.world_size, comm.rank
        if device.type == "cuda":
            lib = _declare_ipc()
            self._lib = lib
            p = ctypes.c_void_p()
            with torch.cuda.device(device):
                hip.check(lib.dtf_ipc_alloc(ctypes.byref(p), self.n * 4), "dtf_ipc_alloc")
                self._own_ptr = p.value
                hb = int(lib.dtf_ipc_handle_bytes())
                buf = ctypes.create_string_buffer(hb)
                if W > 1:
                    hip.check(lib.dtf_ipc_get_handle(p, buf), "hipIpcGetMemHandle")
                handles = self._exchange((bytes(buf.raw), os.getpid()))
                for j in range(W):
                    if j == r:
                        ptr = self._own_ptr
                    else:
                        h, pid = handles[j]
                        assert pid != os.getpid(), "two ranks in one process cannot share IPC memory"
                        q = ctypes.c_void_p()
                        hip.check(lib.dtf_ipc_open(ctypes.create_string_buffer(h, hb), ctypes.byref(q)),
                                  "hipIpcOpenMemHandle")
                        ptr = q.value
                        self._peer_ptrs.append(ptr)
                    self.views[j] = torch.as_tensor(_DevicePtr(ptr, self.n), device=device)
        else:
            base = shm_dir or os.environ.get("DTF_ASYNC_SHM_DIR") or (
                "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
            tag = self._exchange(os.urandom(6).hex() if r == 0 else None)[0]
            paths = [os.path.join(base, "dtf-async-%s-%d.f32" % (tag, j)) for j in range(W)]
            with open(paths[r], "wb") as f:          # the owner creates (zero-filled) its file
                f.truncate(self.n * 4)
            self._barrier()
            for j in range(W):
                self.views[j] = torch.from_file(paths[j], shared=True, size=self.n, dtype=torch.float32)
            self._barrier()
            self._files = [paths[r]]
            for pth in self._files:                  # mapped everywhere: the name can go
                try:
                    os.unlink(pth)
                except OSError:
                    pass

    def _exchange(self, obj):
        if not self.comm.live:
            return [obj]
        out = [None] * self.comm.world_size
        dist.all_gather_object(out, obj, group=self.comm.group)
        return out

    def _barrier(self):
        if self.comm.live:
            dist.barrier(group=self.comm.group)

    def close(self):
        """Every rank must have stopped touching peer memory (callers barrier first)."""
        self.views = []
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                torch.cuda.synchronize(self.device)
                for ptr in self._peer_ptrs:
                    self._lib.dtf_ipc_close(ctypes.c_void_p(ptr))
                self._peer_ptrs = []
                self._barrier()                      # nobody maps my region any more
                if self._own_ptr:
                    self._lib.dtf_ipc_free(ctypes.c_void_p(self._own_ptr))
                    self._own_ptr = None


class AsyncPS:
    """Same ``step()`` surface as :class:`~.ps_runtime.ShardedPS`; ``update_mode = "async"``."""

    update_mode = "async"
    staleness = 0
    family = "one-sided"
    push = "peer-apply"

    def __init__(self, named_params, comm: Comm = None, opt: OptimizerSpec = None, bucket_mb=32.0, wire_dtype=None,
                 tail_mb=0.0, shm_dir=None, hooks=None):
        self.comm = comm or Comm.single()
        self.opt = opt or OptimizerSpec()
        self.named_params = trainable(named_params)
        self.params = [p for _, p in self.named_params]
        self.device = self.params[0].device
        self.is_cuda = self.device.type == "cuda"
        self.wire_dtype = wire_dtype or (torch.bfloat16 if self.is_cuda else torch.float32)
        W, r = self.comm.world_size, self.comm.rank
        self.num_shards = W
        self.hosted = True
        esize = torch.empty((), dtype=self.wire_dtype).element_size()
        decays = [self.opt.decays(n) for n, _ in self.named_params]
        split = self.opt.weight_decay > 0 and not all(decays)
        self.layout = FlatLayout([(n, p.numel(), 0 if (d or not split) else 1)
                                  for (n, p), d in zip(self.named_params, decays)], W,
                                 max(1, int(bucket_mb * (1 << 20) / esize)), int(tail_mb * (1 << 20) / esize))
        L, dev = self.layout, self.device
        self.param_flat = torch.zeros(L.total, dtype=self.wire_dtype, device=dev)
        self.grad_flat = torch.zeros(L.total, dtype=self.wire_dtype, device=dev)
        self.nslots = {"adam": 2, "momentum": 1}.get(self.opt.kind, 1)     # plain SGD keeps a (zero-momentum) slot
        S = L.shard_total
        # region layout: [hyper | master | m | (v)], every part 64-B aligned
        self._off_master = _HYPER * 2
        self._off_m = self._off_master + (S + 15) // 16 * 16
        self._off_v = self._off_m + (S + 15) // 16 * 16
        nfloat = self._off_v + ((S + 15) // 16 * 16 if self.nslots == 2 else 0)
        self.regions = _SharedRegions(self.comm, dev, nfloat, shm_dir)
        v = self.regions.views
        self._hyper = [v[j][:_HYPER] for j in range(W)]
        self._master = [v[j][self._off_master:self._off_master + S] for j in range(W)]
        self._m = [v[j][self._off_m:self._off_m + S] for j in range(W)]
        self._v = [v[j][self._off_v:self._off_v + S] if self.nslots == 2 else None for j in range(W)]
        self.master, self.slot_m = self._master[r], (self._m[r] if self.opt.kind != "sgd" else None)
        self.slot_v = self._v[r]
        self.hyper = self._hyper[r]
        self.hyper[:3].copy_(torch.tensor([self.opt.lr, 1.0, 1.0]))
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev) if self.opt.max_grad_norm > 0 else None
        self.global_step = 0
        self._lr = self.opt.lr

        # -- the chief's initial values go one-sidedly into every owner's master; everyone then reads them back
        self.zero_copy = []
        for slot, p in zip(L.slots, self.params):
            assert bucket_ops.is_dense(p.data), "parameter %s is not dense" % slot.name
            self.param_flat[slot.flat_off:slot.flat_off + slot.numel].copy_(bucket_ops.dense_storage_view(p.data))
        if r == 0:
            for bk in L.buckets:
                for j in range(W):
                    lo, hi = L.shard_range(bk.index, j)
                    self._master[j][bk.shard_off:bk.shard_off + bk.shard].copy_(self.param_flat[lo:hi])
        self._sync_device()
        self.regions._barrier()
        for bk in L.buckets:
            for j in range(W):
                lo, hi = L.shard_range(bk.index, j)
                self.param_flat[lo:hi].copy_(self._master[j][bk.shard_off:bk.shard_off + bk.shard])
        if self.wire_dtype != torch.float32 and r == 0:
            # the master is seeded from the ROUNDED wire values, like ShardedPS (identical start everywhere)
            for bk in L.buckets:
                for j in range(W):
                    lo, hi = L.shard_range(bk.index, j)
                    self._master[j][bk.shard_off:bk.shard_off + bk.shard].copy_(self.param_flat[lo:hi])
        self._sync_device()
        self.regions._barrier()
        for slot, p in zip(L.slots, self.params):
            zc = p.dtype == self.wire_dtype
            if zc:
                p.data = torch.as_strided(self.param_flat, p.shape, p.stride(), slot.flat_off)
            self.zero_copy.append(zc)
        self._unpack_needed = not all(self.zero_copy)
        if self._unpack_needed:
            self._unpack_params()
        self._plans = [bucket_ops.PackPlan([L.slots[i].numel for i in bk.params],
                                           [L.slots[i].flat_off for i in bk.params]) for bk in L.buckets]
        self._closed = False
        # -- hook-driven pipeline (see the module docstring)
        self.comm_stream = torch.cuda.Stream(device=dev) if self.is_cuda else None
        nb = len(L.buckets)
        self._bucket_of = {}
        for bk in L.buckets:
            for i in bk.params:
                self._bucket_of[i] = bk.index
        self._pending = [len(bk.params) for bk in L.buckets]
        self._packed = [False] * nb
        self._applied = [False] * nb
        self._hooks = []
        if hooks is None:
            hooks = os.environ.get("DTF_ASYNC_HOOKS", "1") != "0"
        if hooks:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_grad(i)))

    def _sync_device(self):
        if self.is_cuda:
            torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ step
    def _grads_of(self, b):
        out = []
        for i in self.layout.buckets[b].params:
            p = self.params[i]
            g = p.grad
            if g is None:
                g = torch.zeros_like(p)
                p.grad = g
            elif g.stride() != p.stride() or not bucket_ops.is_dense(g):
                g = torch.empty_like(p).copy_(g)
                p.grad = g
            out.append(g)
        return out

    @contextlib.contextmanager
    def _on_comm_stream(self):
        """Run the body on the second stream, ordered after everything launched so far on the current stream and on
        the weight-gradient side stream (``ops/streams.py``)."""
        if self.comm_stream is None:
            yield
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
        streams_mod.join_into(self.comm_stream, self.device)
        with torch.cuda.stream(self.comm_stream):
            yield

    def _on_grad(self, i):
        b = self._bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._packed[b]:
            with self._on_comm_stream():
                self._pack(b)
                if self.gnorm_sq is None:
                    self._apply(b)

    def _pack(self, b):
        self._plans[b].pack(self._grads_of(b), self.grad_flat, 1.0)           # un-averaged: TF1 async semantics
        self._packed[b] = True

    def _apply(self, b):
        """This worker's gradient of bucket ``b`` onto every owner's slice of it, owners in rotated order."""
        L, W, r, o = self.layout, self.comm.world_size, self.comm.rank, self.opt
        bk = L.buckets[b]
        sl = slice(bk.shard_off, bk.shard_off + bk.shard)
        wd = o.weight_decay if bk.group == 0 else 0.0
        for k in range(W):
            j = (r + k) % W
            lo, hi = L.shard_range(bk.index, j)
            common = dict(grad_scale=1.0, hyper=self._hyper[j], gnorm_sq=self.gnorm_sq, max_norm=o.max_grad_norm)
            if o.kind == "adam":
                optim_ops.adam_apply(self.grad_flat[lo:hi], self._master[j][sl], self._m[j][sl], self._v[j][sl],
                                     self.param_flat[lo:hi], lr=self._lr, beta1=o.beta1, beta2=o.beta2, eps=o.eps,
                                     wd=wd, adamw=o.adamw, bias_correction=o.bias_correction, step=1, **common)
            else:
                mom = o.momentum if o.kind == "momentum" else 0.0
                optim_ops.sgd_momentum_apply(self.grad_flat[lo:hi], self._master[j][sl], self._m[j][sl],
                                             self.param_flat[lo:hi], lr=self._lr, momentum=mom, wd=wd,
                                             nesterov=o.nesterov, **common)
        self._applied[b] = True

    def step(self):
        """Apply THIS worker's gradients to every owner's shard and pull the result -- no other rank involved.
        Buckets whose hooks fired during backward are already packed (and, without clipping, applied) on the second
        stream; whatever is left happens here, then the calling stream joins."""
        L, W, r = self.layout, self.comm.world_size, self.comm.rank
        nb = len(L.buckets)
        if self.is_cuda:
            streams_mod.join_into(torch.cuda.current_stream(self.device), self.device)
        with self._on_comm_stream():
            for b in range(nb - 1, -1, -1):
                if not self._packed[b]:
                    self._pack(b)
            if self.gnorm_sq is not None:
                # each worker clips by the norm of ITS OWN gradient (what clip_by_global_norm does in an async graph)
                reduce_ops.sumsq(self.grad_flat, self.gnorm_sq)
            for b in range(nb - 1, -1, -1):
                if not self._applied[b]:
                    self._apply(b)
            for k in range(W):
                self._hyper[(r + k) % W][2:3].add_(1.0)     # the owner's apply counter (TF global_step); racy by design
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        if self._unpack_needed:
            self._unpack_params()
        for p in self.params:
            p.grad = None
        self._pending = [len(bk.params) for bk in L.buckets]
        self._packed = [False] * nb
        self._applied = [False] * nb
        self.global_step += 1

    def applies_received(self):
        """Number of gradient applies this rank's shard has received from all workers."""
        return int(round(float(self.hyper[2].item()))) - 1

    def _unpack_params(self):
        for bk in self.layout.buckets:
            idx = [i for i in bk.params if not self.zero_copy[i]]
            if idx:
                plan = bucket_ops.PackPlan([self.layout.slots[i].numel for i in idx],
                                           [self.layout.slots[i].flat_off for i in idx])
                plan.unpack(self.param_flat, [self.params[i].data for i in idx])

    # ------------------------------------------------------------- controls
    def set_lr(self, lr):
        """The learning rate lives with the shard: every rank sets the one it hosts."""
        if lr != self._lr:
            self._lr = float(lr)
            self.hyper[0:1].fill_(self._lr)

    @property
    def lr(self):
        return self._lr

    def pull(self):
        """Refresh the local parameters from every owner's master (e.g. before evaluation)."""
        L, W = self.layout, self.comm.world_size
        for bk in L.buckets:
            for j in range(W):
                lo, hi = L.shard_range(bk.index, j)
                self.param_flat[lo:hi].copy_(self._master[j][bk.shard_off:bk.shard_off + bk.shard])
        if self._unpack_needed:
            self._unpack_params()

    def flush(self):
        self._sync_device()

    def finish(self):
        """End of training: wait for every worker's last apply, then adopt the final values."""
        self._sync_device()
        self.regions._barrier()
        self.pull()
        self._sync_device()

    def full_master(self):
        L, W = self.layout, self.comm.world_size
        out = torch.zeros(L.total, dtype=torch.float32, device=self.device)
        for bk in L.buckets:
            for j in range(W):
                lo, hi = L.shard_range(bk.index, j)
                out[lo:hi].copy_(self._master[j][bk.shard_off:bk.shard_off + bk.shard])
        return out

    def param_from_flat(self, flat, index):
        s = self.layout.slots[index]
        p = self.params[index]
        return torch.as_strided(flat, p.shape, p.stride(), s.flat_off)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def close(self):
        """Detach the model from the shared regions and release them (collective: every rank calls it)."""
        if self._closed:
            return
        self._closed = True
        self.remove_hooks()
        self._sync_device()
        self.regions._barrier()
        # own-shard views die with the regions: keep private copies for checkpoints written afterwards
        self.master = self.master.clone()
        self.slot_m = None if self.slot_m is None else self.slot_m.clone()
        self.slot_v = None if self.slot_v is None else self.slot_v.clone()
        self.hyper = self.hyper.clone()
        self._hyper = self._master = self._m = self._v = []
        self.regions.close()
