"""Sharded parameter-server runtime (new layer N2; replaces externals X3-X5: the
PS-hosted variables, ``Apply*`` optimizer ops and sync/async update modes of the
TF1 runtime that the reference's jobs run).

Training step of rank r of W (colocated placement -- every GPU rank is a worker
*and* hosts 1/W of the PS state in its HBM)::

    backward ──grad hooks──▶ K5 pack bucket b (grads -> wire-dtype flat, x 1/W)
                             push:  RCCL reduce_scatter(bucket b)      [xGMI]
                                    (or all_to_all + W-way sum fused in K1/K2)
                             K1/K2 fused shard apply on the owned slice
                             pull:  RCCL all_gather(bucket b)          [xGMI]
    next forward reads the pulled wire-dtype parameters in place: the model's
    parameters are views into the flat pull buffer, so there is no unpack.

Everything after the hook runs on a side stream and overlaps with the rest of
backward; ``step()`` joins the streams.  No host synchronisation in the step.

Update modes:
  ``sync``  -- push/apply/pull every step before the next forward
               (TF ``SyncReplicasOptimizer`` semantics; deterministic).
  ``stale`` -- bounded staleness 1: step t's push/apply/pull overlaps with the
               whole forward+backward of step t+1, which still runs on the
               parameters from before update t (the collective counterpart of
               TF1's default asynchronous PS update; deterministic).
True any-order asynchronous updates exist in the dedicated placement
(``parallel/dedicated_ps.py``), where PS tasks are separate processes.
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass

import torch

from ..ops import bucket as bucket_ops
from ..ops import optim as optim_ops
from ..ops import reduce as reduce_ops
from .comm import Comm
from .layout import FlatLayout


_TRACE = os.environ.get("DTF_TRACE", "0") == "1"


@contextlib.contextmanager
def trace_range(name):
    """roctx range (``torch.cuda.nvtx`` is roctx on ROCm) around a PS pipeline stage when
    ``DTF_TRACE=1``: shows pack / push / apply / pull in rocprofv3 ``--marker-trace`` timelines."""
    if _TRACE and torch.cuda.is_available():
        torch.cuda.nvtx.range_push(name)
        try:
            yield
        finally:
            torch.cuda.nvtx.range_pop()
    else:
        yield


@dataclass
class OptimizerSpec:
    kind: str = "momentum"        # sgd | momentum | adam
    lr: float = 0.1
    momentum: float = 0.9
    nesterov: bool = False
    weight_decay: float = 0.0
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    adamw: bool = True
    bias_correction: bool = True
    max_grad_norm: float = 0.0    # 0 = no clipping

    @property
    def slots(self):
        """TF-style optimizer slot names (checkpoint naming)."""
        if self.kind == "adam":
            return ("Adam", "Adam_1")
        if self.kind == "momentum":
            return ("Momentum",)
        return ()


def trainable(named_params):
    """Dense variables that go through the PS buckets: trainable parameters, minus the ones a
    module manages itself (``dtf_skip_ps``, e.g. the autograd anchor of ShardedEmbedding)."""
    return [(n, p) for n, p in named_params if p.requires_grad and not getattr(p, "dtf_skip_ps", False)]


class ShardedPS:
    def __init__(self, named_params, comm: Comm = None, opt: OptimizerSpec = None, bucket_mb=32.0,
                 wire_dtype=None, update_mode="sync", push="reduce_scatter", overlap=True,
                 broadcast_init=True, tail_mb=4.0):
        self.comm = comm or Comm.single()
        self.opt = opt or OptimizerSpec()
        self.named_params = trainable(named_params)
        assert self.named_params, "no trainable parameters"
        self.params = [p for _, p in self.named_params]
        self.device = self.params[0].device
        self.is_cuda = self.device.type == "cuda"
        self.wire_dtype = wire_dtype or (torch.bfloat16 if self.is_cuda else torch.float32)
        assert update_mode in ("sync", "stale"), update_mode
        assert push in ("reduce_scatter", "all_to_all"), push
        self.update_mode, self.push = update_mode, push
        W, r = self.comm.world_size, self.comm.rank
        self._dist = self.comm.live          # collectives are real (W > 1, or a forced 1-rank group)
        esize = torch.empty((), dtype=self.wire_dtype).element_size()
        # tail_mb: size cap of the bucket holding the FIRST layers -- the only one whose push/apply/pull is
        # exposed after backward (parallel/layout.py)
        self.layout = FlatLayout([(n, p.numel()) for n, p in self.named_params], W,
                                 max(1, int(bucket_mb * (1 << 20) / esize)), int(tail_mb * (1 << 20) / esize))
        L = self.layout
        dev = self.device
        self.param_flat = torch.zeros(L.total, dtype=self.wire_dtype, device=dev)
        self.grad_flat = torch.zeros(L.total, dtype=self.wire_dtype, device=dev)
        self.master = torch.zeros(L.shard_total, dtype=torch.float32, device=dev)
        self.slot_m = torch.zeros(L.shard_total, dtype=torch.float32, device=dev) \
            if self.opt.kind in ("momentum", "adam") else None
        self.slot_v = torch.zeros(L.shard_total, dtype=torch.float32, device=dev) \
            if self.opt.kind == "adam" else None
        if self._dist:
            self.grad_shard = torch.zeros(L.total if push == "all_to_all" else L.shard_total,
                                          dtype=self.wire_dtype, device=dev)
            self.shard_out = torch.zeros(L.shard_total, dtype=self.wire_dtype, device=dev)
        else:
            self.grad_shard = self.shard_out = None
        self.pull_staging = torch.zeros(L.total, dtype=self.wire_dtype, device=dev) \
            if update_mode == "stale" else None
        # hyper-parameters live in device memory so a captured hipGraph replays
        # with new values: [lr, extra grad scale, step]
        self.hyper = torch.tensor([self.opt.lr, 1.0, 1.0], dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev) if self.opt.max_grad_norm > 0 else None
        self.global_step = 0
        self._lr = self.opt.lr

        # -- bind model parameters to the flat pull buffer (zero-copy pull) ---
        if broadcast_init and self._dist:
            for p in self.params:
                # the dense 1-D view of the storage: NHWC (channels_last) conv filters are not "contiguous" in the
                # default sense, and not every backend accepts that
                self.comm.broadcast(bucket_ops.dense_storage_view(p.data), 0)
        self.zero_copy = []
        for slot, p in zip(L.slots, self.params):
            assert bucket_ops.is_dense(p.data), "parameter %s is not dense" % slot.name
            region = self.param_flat[slot.flat_off:slot.flat_off + slot.numel]
            region.copy_(bucket_ops.dense_storage_view(p.data))
            zc = p.dtype == self.wire_dtype
            if zc:
                p.data = torch.as_strided(self.param_flat, p.shape, p.stride(), slot.flat_off)
            self.zero_copy.append(zc)
        # fp32 master copy of the owned slices.  With a bf16 wire the master is
        # seeded from the (rounded) wire values, identical on every rank.
        for b in L.buckets:
            lo, hi = L.shard_range(b.index, r)
            self.master[b.shard_off:b.shard_off + b.shard].copy_(self.param_flat[lo:hi])
        self._plans = []
        for b in L.buckets:
            numels = [L.slots[i].numel for i in b.params]
            offs = [L.slots[i].flat_off for i in b.params]
            self._plans.append(bucket_ops.PackPlan(numels, offs))
        self._unpack_needed = not all(self.zero_copy)

        # -- overlap machinery ------------------------------------------------
        self.overlap = bool(overlap) and update_mode == "sync"
        self.comm_stream = torch.cuda.Stream(device=dev) if self.is_cuda else None
        self._pending = [len(b.params) for b in L.buckets]
        self._launched = [False] * len(L.buckets)
        self._slot_bucket = {}
        self._hooks = []
        for slot, p in zip(L.slots, self.params):
            self._slot_bucket[id(p)] = slot.bucket
            if self.overlap:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._inflight = False

    # ------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        b = self._slot_bucket[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b] and self.gnorm_sq is None:
            self._launch_bucket(b)

    @contextlib.contextmanager
    def _on_comm_stream(self):
        if self.comm_stream is None:
            yield
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.comm_stream):
            yield

    def _grads_of(self, b):
        out = []
        for i in self.layout.buckets[b].params:
            p = self.params[i]
            g = p.grad
            if g is None:
                g = torch.zeros_like(p)     # unused parameter this step
                p.grad = g
            elif g.stride() != p.stride() or not bucket_ops.is_dense(g):
                g = torch.empty_like(p).copy_(g)
                p.grad = g
            out.append(g)
        return out

    # ------------------------------------------------------------- pipeline
    def _pack(self, b):
        self._plans[b].pack(self._grads_of(b), self.grad_flat, 1.0 / self.comm.world_size)

    def _push(self, b):
        L, W = self.layout, self.comm.world_size
        bk = L.buckets[b]
        if not self._dist:
            return self.grad_flat[bk.start:bk.start + bk.size], 1, 0
        full = self.grad_flat[bk.start:bk.start + bk.size]
        if self.push == "all_to_all":
            recv = self.grad_shard[bk.start:bk.start + bk.size]
            self.comm.all_to_all(recv, full)
            return recv, W, bk.shard
        out = self.grad_shard[bk.shard_off:bk.shard_off + bk.shard]
        self.comm.reduce_scatter(out, full)
        return out, 1, 0

    def _apply(self, b, g, nsrc, src_stride):
        L, W, r = self.layout, self.comm.world_size, self.comm.rank
        bk = L.buckets[b]
        sl = slice(bk.shard_off, bk.shard_off + bk.shard)
        if not self._dist:
            dst = self.pull_staging if self.update_mode == "stale" else self.param_flat
            p_out = dst[bk.start:bk.start + bk.size]
        else:
            p_out = self.shard_out[sl]
        o = self.opt
        common = dict(grad_scale=1.0, nsrc=nsrc, src_stride=src_stride, hyper=self.hyper,
                      gnorm_sq=self.gnorm_sq, max_norm=o.max_grad_norm)
        if o.kind == "adam":
            optim_ops.adam_apply(g, self.master[sl], self.slot_m[sl], self.slot_v[sl], p_out, lr=self._lr,
                                 beta1=o.beta1, beta2=o.beta2, eps=o.eps, wd=o.weight_decay, adamw=o.adamw,
                                 bias_correction=o.bias_correction, step=self.global_step + 1, **common)
        else:
            mom = o.momentum if o.kind == "momentum" else 0.0
            m = self.slot_m[sl] if self.slot_m is not None else self._scratch_m(bk.shard)
            optim_ops.sgd_momentum_apply(g, self.master[sl], m, p_out, lr=self._lr, momentum=mom,
                                         wd=o.weight_decay, nesterov=o.nesterov, **common)

    def _scratch_m(self, n):
        # plain SGD: momentum 0 makes m == g; a scratch slot keeps one kernel
        if getattr(self, "_m_scratch", None) is None or self._m_scratch.numel() < n:
            self._m_scratch = torch.zeros(n, dtype=torch.float32, device=self.device)
        return self._m_scratch[:n]

    def _pull(self, b):
        L, W = self.layout, self.comm.world_size
        if not self._dist:
            return
        bk = L.buckets[b]
        dst = self.pull_staging if self.update_mode == "stale" else self.param_flat
        self.comm.all_gather(dst[bk.start:bk.start + bk.size],
                             self.shard_out[bk.shard_off:bk.shard_off + bk.shard])

    def _launch_bucket(self, b):
        """pack -> push -> apply -> pull of one bucket on the comm stream."""
        self._launched[b] = True
        with self._on_comm_stream():
            with trace_range("ps/pack/b%d" % b):
                self._pack(b)
            with trace_range("ps/push/b%d" % b):
                g, nsrc, stride = self._push(b)
            with trace_range("ps/apply/b%d" % b):
                self._apply(b, g, nsrc, stride)
            with trace_range("ps/pull/b%d" % b):
                self._pull(b)

    # ------------------------------------------------------------------ step
    def step(self):
        """Finish the update of the step whose backward just ran."""
        nb = len(self.layout.buckets)
        if self.update_mode == "stale":
            self._step_stale()
        elif self.gnorm_sq is not None:
            self._step_clipped()
        else:
            for b in range(nb):
                if not self._launched[b]:
                    self._launch_bucket(b)
            with self._on_comm_stream():
                self.hyper[2:3].add_(1.0)
            self._join()
        self._finish_step()

    def _join(self):
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)

    def _step_clipped(self):
        """Global-norm clipping needs every reduced shard before any apply."""
        nb = len(self.layout.buckets)
        with self._on_comm_stream():
            pushed = []
            for b in range(nb):
                self._pack(b)
                pushed.append(self._push(b))
            first = True
            for b, (g, nsrc, stride) in enumerate(pushed):
                bk = self.layout.buckets[b]
                if nsrc > 1:   # all_to_all slabs: reduce first so the norm sees the sum
                    acc = g[:bk.shard].float()
                    for s in range(1, nsrc):
                        acc += g[s * stride:s * stride + bk.shard].float()
                    g = acc.to(self.wire_dtype)
                    pushed[b] = (g, 1, 0)
                n = bk.shard if self._dist else bk.size
                reduce_ops.sumsq(g[:n], self.gnorm_sq, accumulate=not first)
                first = False
            self.comm.all_reduce(self.gnorm_sq)
            for b, (g, nsrc, stride) in enumerate(pushed):
                self._apply(b, g, nsrc, stride)
                self._pull(b)
            self.hyper[2:3].add_(1.0)
        self._join()

    def _step_stale(self):
        nb = len(self.layout.buckets)
        # 1. the previous step's pipeline must be done; adopt its parameters
        self._join()
        if self._inflight:
            self.param_flat.copy_(self.pull_staging)
        # 2. stage this step's gradients, then run the pipeline behind the
        #    next forward/backward
        for b in range(nb):
            self._pack(b)
        with self._on_comm_stream():
            for b in range(nb):
                g, nsrc, stride = self._push(b)
                self._apply(b, g, nsrc, stride)
                self._pull(b)
            self.hyper[2:3].add_(1.0)
        self._inflight = True

    def _finish_step(self):
        if self._unpack_needed:
            self._unpack_params()
        for p in self.params:
            p.grad = None
        self._pending = [len(b.params) for b in self.layout.buckets]
        self._launched = [False] * len(self.layout.buckets)
        self.global_step += 1

    def flush(self):
        """Drain a stale-mode pipeline (end of training / before checkpoint)."""
        self._join()
        if self.update_mode == "stale" and self._inflight:
            self.param_flat.copy_(self.pull_staging)
            self._inflight = False
            if self._unpack_needed:
                self._unpack_params()

    def _unpack_params(self):
        # parameters whose dtype differs from the wire dtype (fp32 model, bf16 wire)
        for b, bk in enumerate(self.layout.buckets):
            idx = [i for i in bk.params if not self.zero_copy[i]]
            if not idx:
                continue
            plan = bucket_ops.PackPlan([self.layout.slots[i].numel for i in idx],
                                       [self.layout.slots[i].flat_off for i in idx])
            plan.unpack(self.param_flat, [self.params[i].data for i in idx])

    # ------------------------------------------------------------- controls
    def set_lr(self, lr):
        if lr != self._lr:
            self._lr = float(lr)
            self.hyper[0:1].fill_(self._lr)

    @property
    def lr(self):
        return self._lr

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def full_master(self):
        """All-gather the fp32 master weights (tests / export): flat [total]."""
        L, W = self.layout, self.comm.world_size
        self.flush()
        out = torch.zeros(L.total, dtype=torch.float32, device=self.device)
        for bk in L.buckets:
            sl = self.master[bk.shard_off:bk.shard_off + bk.shard].contiguous()
            self.comm.all_gather(out[bk.start:bk.start + bk.size], sl)
        return out

    def param_from_flat(self, flat, index):
        s = self.layout.slots[index]
        p = self.params[index]
        return torch.as_strided(flat, p.shape, p.stride(), s.flat_off)
