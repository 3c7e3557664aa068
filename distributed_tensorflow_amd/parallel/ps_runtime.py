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
  ``stale`` -- bounded staleness ``s`` (default 1): step t's push/apply/pull overlaps
               with the whole forward+backward of steps t+1 .. t+s, which still run on
               the parameters from before update t (stale-synchronous-parallel: the
               collective counterpart of TF1's asynchronous PS update; deterministic).
True any-order asynchronous updates: ``parallel/async_ps.py`` (colocated GPU ranks,
one-sided shard apply over xGMI peer mappings, no collectives and no barrier) and
``parallel/dedicated_ps.py`` (PS tasks as separate processes).

Shard groups: ``num_shards`` is the job's ``detail.ps`` (the reference creates that many
PS tasks which own the variables, ``handlers/train_handler.py:96-112``).  ``P >= W``
spreads the state uniformly over all W ranks (push = reduce-scatter, pull = all-gather,
every xGMI link busy).  ``P < W`` is the classic "few parameter servers" topology: ranks
0..P-1 host the state, push = ``reduce`` of slice j to rank j, pull = ``broadcast`` of
slice j from rank j.

Gradient clipping (``max_grad_norm``): packs and pushes still run from the backward
hooks; only sum-of-squares -> scalar all-reduce -> apply -> pull wait for the end of
backward, and they then run in FORWARD layer order (the bucket of the first layers first).

Pull prefetch (``attach(model)``): every bucket's pull records an event; forward pre-hooks
make a layer wait for the event of ITS bucket only, so the next forward starts as soon as
the first layers' parameters have arrived and the remaining all-gathers (BERT: ~220 MB
after the clip all-reduce) run under it -- SURVEY 5.8 "forward-layer order for pull
prefetch".  Without ``attach`` the next forward waits for the whole comm stream.
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass

import torch

from ..ops import bucket as bucket_ops
from ..ops import optim as optim_ops
from ..ops import reduce as reduce_ops
from ..ops import streams as streams_mod
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
    no_decay: tuple = ()          # name substrings of variables exempt from weight decay (BERT: LayerNorm, bias)

    @property
    def slots(self):
        """TF-style optimizer slot names (checkpoint naming)."""
        if self.kind == "adam":
            return ("Adam", "Adam_1")
        if self.kind == "momentum":
            return ("Momentum",)
        return ()

    def decays(self, name):
        return not any(pat in name for pat in self.no_decay)


def trainable(named_params):
    """Dense variables that go through the PS buckets: trainable parameters, minus the ones a
    module manages itself (``dtf_skip_ps``, e.g. the autograd anchor of ShardedEmbedding)."""
    return [(n, p) for n, p in named_params if p.requires_grad and not getattr(p, "dtf_skip_ps", False)]


class ShardedPS:
    def __init__(self, named_params, comm: Comm = None, opt: OptimizerSpec = None, bucket_mb=32.0,
                 wire_dtype=None, update_mode="sync", push="reduce_scatter", overlap=True,
                 broadcast_init=True, tail_mb=4.0, num_shards=None, staleness=1):
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
        self.update_mode = update_mode
        self.staleness = max(1, int(staleness)) if update_mode == "stale" else 0
        W, r = self.comm.world_size, self.comm.rank
        # shard groups (detail.ps): P >= W -> uniform over the W ranks; P < W -> ranks 0..P-1 host the state
        P = W if (not num_shards or int(num_shards) >= W) else max(1, int(num_shards))
        self.num_shards = P
        self.family = "scatter" if P == W else "reduce"
        self.hosted = r < P
        self.push = push if self.family == "scatter" else "reduce"
        self._dist = self.comm.live          # collectives are real (W > 1, or a forced 1-rank group)
        esize = torch.empty((), dtype=self.wire_dtype).element_size()
        # tail_mb: size cap of the bucket holding the FIRST layers -- the only one whose push/apply/pull is
        # exposed after backward (parallel/layout.py)
        decays = [self.opt.decays(n) for n, _ in self.named_params]
        split_groups = self.opt.weight_decay > 0 and not all(decays)
        self.layout = FlatLayout([(n, p.numel(), 0 if (d or not split_groups) else 1)
                                  for (n, p), d in zip(self.named_params, decays)], P,
                                 max(1, int(bucket_mb * (1 << 20) / esize)), int(tail_mb * (1 << 20) / esize))
        L = self.layout
        dev = self.device
        self.param_flat = torch.zeros(L.total, dtype=self.wire_dtype, device=dev)
        n_state = L.shard_total if self.hosted else 0
        self.master = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.slot_m = torch.zeros(n_state, dtype=torch.float32, device=dev) \
            if self.opt.kind in ("momentum", "adam") else None
        self.slot_v = torch.zeros(n_state, dtype=torch.float32, device=dev) \
            if self.opt.kind == "adam" else None
        # gradient staging: one flat buffer, or a ring of `staleness` of them (a stale pipeline reads its
        # gradients while later steps pack theirs)
        nring = max(1, self.staleness)
        self._grad_ring = [torch.zeros(L.total, dtype=self.wire_dtype, device=dev) for _ in range(nring)]
        self.grad_flat = self._grad_ring[0]
        if self._dist and self.family == "scatter":
            self.grad_shard = torch.zeros(L.total if push == "all_to_all" else L.shard_total,
                                          dtype=self.wire_dtype, device=dev)
            self.shard_out = torch.zeros(L.shard_total, dtype=self.wire_dtype, device=dev)
        else:
            self.grad_shard = self.shard_out = None
        self._staging = [torch.zeros(L.total, dtype=self.wire_dtype, device=dev) for _ in range(self.staleness)]
        self.pull_staging = self._staging[0] if self._staging else None
        # hyper-parameters live in device memory so a captured hipGraph replays
        # with new values: [lr, extra grad scale, step]
        self.hyper = torch.tensor([self.opt.lr, 1.0, 1.0], dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev) if self.opt.max_grad_norm > 0 else None
        self.global_step = 0
        self._lr = self.opt.lr

        # -- bind model parameters to the flat pull buffer (zero-copy pull) ---
        self.zero_copy = []
        for slot, p in zip(L.slots, self.params):
            assert bucket_ops.is_dense(p.data), "parameter %s is not dense" % slot.name
            self.param_flat[slot.flat_off:slot.flat_off + slot.numel].copy_(bucket_ops.dense_storage_view(p.data))
        if broadcast_init and self._dist:
            # the chief's initial values, one collective per BUCKET of the flat buffer (not one per variable:
            # 161 broadcasts for ResNet-50, ~200 for BERT)
            for bk in L.buckets:
                self.comm.broadcast(self.param_flat[bk.start:bk.start + bk.size], 0)
        for slot, p in zip(L.slots, self.params):
            zc = p.dtype == self.wire_dtype
            if zc:
                p.data = torch.as_strided(self.param_flat, p.shape, p.stride(), slot.flat_off)
            self.zero_copy.append(zc)
        self._unpack_needed = not all(self.zero_copy)
        if self._unpack_needed and broadcast_init and self._dist:
            self._unpack_params()
        # fp32 master copy of the owned slices.  With a bf16 wire the master is
        # seeded from the (rounded) wire values, identical on every rank.
        if self.hosted:
            for b in L.buckets:
                lo, hi = L.shard_range(b.index, r)
                self.master[b.shard_off:b.shard_off + b.shard].copy_(self.param_flat[lo:hi])
        self._plans = []
        for b in L.buckets:
            numels = [L.slots[i].numel for i in b.params]
            offs = [L.slots[i].flat_off for i in b.params]
            self._plans.append(bucket_ops.PackPlan(numels, offs))

        # -- overlap machinery ------------------------------------------------
        self.overlap = bool(overlap) and update_mode == "sync"
        self.comm_stream = torch.cuda.Stream(device=dev) if self.is_cuda else None
        self._pending = [len(b.params) for b in L.buckets]
        self._launched = [False] * len(L.buckets)
        self._pushed = [None] * len(L.buckets)
        self._slot_bucket = {}
        self._hooks = []
        for slot, p in zip(L.slots, self.params):
            self._slot_bucket[id(p)] = slot.bucket
            if self.overlap:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._inflight = [None] * self.staleness      # per ring slot: event of the pipeline in flight (or True)
        self._last_slot = None
        # stale mode: the packs run from the hooks too (on the comm stream, behind the pipeline still in flight)
        if bool(overlap) and update_mode == "stale":
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_stale))
        # -- pull prefetch (see attach) ---------------------------------------
        nb = len(L.buckets)
        self._pull_events = [torch.cuda.Event() for _ in range(nb)] if self.is_cuda else [None] * nb
        self._pull_pending = [False] * nb     # pulled on the comm stream, not yet waited for by the compute stream
        self._prefetch = False                # per-bucket waits are live (attach + one learning step done)
        self._grad_keep = []                  # gradients of earlier steps whose packs this stream has not waited for yet
        self.prefetch_waits = 0               # per-bucket event waits issued so far (tests / diagnostics)
        self._attached = None

    # ------------------------------------------------------------------ hooks
    def _on_grad(self, p):
        b = self._slot_bucket[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            if self.gnorm_sq is None:
                self._launch_bucket(b)
            else:
                self._launch_push(b)

    def _on_grad_stale(self, p):
        b = self._slot_bucket[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            self._launched[b] = True
            slot = self.global_step % self.staleness
            with self._on_comm_stream():
                with trace_range("ps/pack/b%d" % b):
                    self._pack(b, self._grad_ring[slot])

    # ------------------------------------------------------- pull prefetch
    def attach(self, model, prefetch=None):
        """Forward-order pull prefetch: instead of joining the whole comm stream before the next forward, every
        module that owns parameters waits (forward pre-hook) for the pull event of the buckets holding THEM.

        Parameters that are read outside their owner's ``forward`` (BERT's root module reads
        ``word_embeddings.weight`` functionally, twice) cannot be caught by the owner's hook.  They are found by
        observation: the first step after ``attach`` still joins the whole stream and records which owner modules
        were called; the buckets of every owner that was NOT called are waited for at the root's pre-hook, i.e.
        before anything runs.  ``DTF_PULL_PREFETCH=0`` (or ``prefetch=False``) keeps the global join."""
        if prefetch is None:
            prefetch = os.environ.get("DTF_PULL_PREFETCH", "1") != "0"
        if not prefetch or not self.is_cuda or self._unpack_needed or self.update_mode != "sync" or not self._dist:
            return self         # nothing to overlap (no comm stream work to wait for / parameters need an unpack)
        bucket_of = {id(p): self._slot_bucket[id(p)] for p in self.params}
        owners = {}
        for m in model.modules():
            bs = sorted({bucket_of[id(p)] for p in m._parameters.values() if p is not None and id(p) in bucket_of})
            if bs:
                owners[m] = bs
        st = self._attached = {"owners": owners, "called": set(), "root": model, "root_buckets": None,
                               "learning": True, "handles": []}

        def owner_pre(m, args):
            if st["learning"]:
                st["called"].add(m)
            elif self._prefetch:
                self._wait_buckets(owners[m])

        def root_pre(m, args):
            if self._prefetch and st["root_buckets"]:
                self._wait_buckets(st["root_buckets"])

        def root_post(m, args, out):
            if st["learning"]:
                never = [b for mod, bs in owners.items() if mod not in st["called"] for b in bs]
                st["root_buckets"] = sorted(set(never))
                st["learning"] = False
            elif self._prefetch:
                self._wait_buckets(range(len(self.layout.buckets)))     # nothing may stay un-waited past forward

        st["handles"].append(model.register_forward_pre_hook(root_pre))
        for m in owners:
            if m is not model:
                st["handles"].append(m.register_forward_pre_hook(owner_pre))
            else:
                # the root owns parameters itself: they belong to the root set
                st["called"].discard(m)
        st["handles"].append(model.register_forward_hook(root_post))
        return self

    def _wait_buckets(self, buckets):
        cur = None
        for b in buckets:
            if self._pull_pending[b]:
                if cur is None:
                    cur = torch.cuda.current_stream(self.device)
                cur.wait_event(self._pull_events[b])
                self._pull_pending[b] = False
                self.prefetch_waits += 1

    def _pulled(self, b):
        """Called on the comm stream right after bucket ``b``'s pull was enqueued."""
        if self._pull_events[b] is not None:
            self._pull_events[b].record(torch.cuda.current_stream(self.device))
            self._pull_pending[b] = True

    @contextlib.contextmanager
    def _on_comm_stream(self):
        if self.comm_stream is None:
            yield
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
        streams_mod.join_into(self.comm_stream, self.device)        # weight gradients computed on their side stream
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
    def _pack(self, b, grad_flat=None):
        self._plans[b].pack(self._grads_of(b), self.grad_flat if grad_flat is None else grad_flat,
                            1.0 / self.comm.world_size)

    def _push(self, b, grad_flat=None):
        """-> (reduced gradient of this rank's slice, nsrc, src_stride), or None when this rank hosts no
        state of the bucket (``P < W``)."""
        L, W = self.layout, self.comm.world_size
        bk = L.buckets[b]
        gf = self.grad_flat if grad_flat is None else grad_flat
        full = gf[bk.start:bk.start + bk.size]
        if not self._dist:
            return full, 1, 0
        if self.family == "reduce":
            # worker -> PS j: one rooted reduce per shard group, in place in the staging bucket
            for j in range(self.num_shards):
                self.comm.reduce(full[j * bk.shard:(j + 1) * bk.shard], j)
            r = self.comm.rank
            return (full[r * bk.shard:(r + 1) * bk.shard], 1, 0) if self.hosted else None
        if self.push == "all_to_all":
            recv = self.grad_shard[bk.start:bk.start + bk.size]
            self.comm.all_to_all(recv, full)
            return recv, W, bk.shard
        out = self.grad_shard[bk.shard_off:bk.shard_off + bk.shard]
        self.comm.reduce_scatter(out, full)
        return out, 1, 0

    def _apply(self, b, pushed, dst=None):
        if pushed is None:
            return
        g, nsrc, src_stride = pushed
        L, r = self.layout, self.comm.rank
        bk = L.buckets[b]
        sl = slice(bk.shard_off, bk.shard_off + bk.shard)
        dst = self.param_flat if dst is None else dst
        if not self._dist:
            p_out = dst[bk.start:bk.start + bk.size]
        elif self.family == "reduce":
            lo, hi = L.shard_range(b, r)
            p_out = dst[lo:hi]
        else:
            p_out = self.shard_out[sl]
        o = self.opt
        wd = o.weight_decay if bk.group == 0 else 0.0
        common = dict(grad_scale=1.0, nsrc=nsrc, src_stride=src_stride, hyper=self.hyper,
                      gnorm_sq=self.gnorm_sq, max_norm=o.max_grad_norm)
        if o.kind == "adam":
            optim_ops.adam_apply(g, self.master[sl], self.slot_m[sl], self.slot_v[sl], p_out, lr=self._lr,
                                 beta1=o.beta1, beta2=o.beta2, eps=o.eps, wd=wd, adamw=o.adamw,
                                 bias_correction=o.bias_correction, step=self.global_step + 1, **common)
        else:
            mom = o.momentum if o.kind == "momentum" else 0.0
            m = self.slot_m[sl] if self.slot_m is not None else self._scratch_m(bk.shard)
            optim_ops.sgd_momentum_apply(g, self.master[sl], m, p_out, lr=self._lr, momentum=mom,
                                         wd=wd, nesterov=o.nesterov, **common)

    def _scratch_m(self, n):
        # plain SGD: momentum 0 makes m == g; a scratch slot keeps one kernel
        if getattr(self, "_m_scratch", None) is None or self._m_scratch.numel() < n:
            self._m_scratch = torch.zeros(n, dtype=torch.float32, device=self.device)
        return self._m_scratch[:n]

    def _pull(self, b, dst=None):
        L = self.layout
        if not self._dist:
            return
        bk = L.buckets[b]
        dst = self.param_flat if dst is None else dst
        if self.family == "reduce":
            # PS j -> every worker: one broadcast per shard group, in place
            for j in range(self.num_shards):
                lo, hi = L.shard_range(b, j)
                self.comm.broadcast(dst[lo:hi], j)
            return
        self.comm.all_gather(dst[bk.start:bk.start + bk.size],
                             self.shard_out[bk.shard_off:bk.shard_off + bk.shard])

    def _launch_bucket(self, b):
        """pack -> push -> apply -> pull of one bucket on the comm stream."""
        self._launched[b] = True
        with self._on_comm_stream():
            with trace_range("ps/pack/b%d" % b):
                self._pack(b)
            with trace_range("ps/push/b%d" % b):
                pushed = self._push(b)
            with trace_range("ps/apply/b%d" % b):
                self._apply(b, pushed)
            with trace_range("ps/pull/b%d" % b):
                self._pull(b)
                self._pulled(b)

    def _launch_push(self, b):
        """Clipped path, from the hooks: pack -> push only; apply waits for the global norm."""
        self._launched[b] = True
        with self._on_comm_stream():
            with trace_range("ps/pack/b%d" % b):
                self._pack(b)
            with trace_range("ps/push/b%d" % b):
                self._pushed[b] = self._push(b)

    # ------------------------------------------------------------------ step
    def step(self):
        """Finish the update of the step whose backward just ran."""
        nb = len(self.layout.buckets)
        if self.is_cuda:
            # the compute stream rejoins the weight-gradient side stream (ops/streams.py): everything that follows on
            # it -- packs of a stale step, the next forward, the end of a hipGraph capture -- is ordered after it
            streams_mod.join_into(torch.cuda.current_stream(self.device), self.device)
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
            self._join_or_prefetch()
        self._finish_step()

    def _join(self):
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        self._pull_pending = [False] * len(self._pull_pending)

    def _join_or_prefetch(self):
        """End of a sync step: either the next forward waits per bucket (prefetch live), or for everything."""
        st = self._attached
        if st is not None and not st["learning"] and st["root_buckets"] is not None \
                and not torch.cuda.is_current_stream_capturing():
            self._prefetch = True
            return
        self._join()

    def disable_prefetch(self):
        """Before a hipGraph capture: a captured step must rejoin the comm stream before the capture ends (and a
        replay runs no Python hooks), so the per-bucket waits are switched off and everything pending is joined."""
        self._prefetch = False
        if self._attached is not None:
            for h in self._attached["handles"]:
                h.remove()
            self._attached = None
        self._join()

    def _step_clipped(self):
        """Global-norm clipping needs every reduced shard before any apply: the packs and pushes already ran
        from the hooks (behind backward); what is left is sum-of-squares of the reduced slices (the W-way slab
        sum of an all-to-all push fused in), one scalar all-reduce, then apply + pull per bucket."""
        nb = len(self.layout.buckets)
        for b in range(nb):
            if not self._launched[b]:
                self._launch_push(b)
        with self._on_comm_stream():
            first = True
            for b in range(nb):
                pushed = self._pushed[b]
                if pushed is None:
                    continue
                g, nsrc, stride = pushed
                bk = self.layout.buckets[b]
                n = bk.shard if self._dist else bk.size
                reduce_ops.sumsq(g, self.gnorm_sq, accumulate=not first, n=n, nsrc=nsrc, src_stride=stride)
                first = False
            if first:
                self.gnorm_sq.zero_()       # this rank hosts nothing (P < W): contributes 0 to the norm
            self.comm.all_reduce(self.gnorm_sq)
            # forward layer order (the last bucket holds the first layers): with the prefetch hooks the next
            # forward starts after the FIRST of these pulls and runs under the others
            for b in reversed(range(nb)):
                with trace_range("ps/apply/b%d" % b):
                    self._apply(b, self._pushed[b])
                with trace_range("ps/pull/b%d" % b):
                    self._pull(b)
                    self._pulled(b)
            self.hyper[2:3].add_(1.0)
        self._pushed = [None] * nb
        self._join_or_prefetch()

    def _wait_slot(self, slot):
        ev = self._inflight[slot]
        if ev is None:
            return False
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return True

    def _step_stale(self):
        nb = len(self.layout.buckets)
        slot = self.global_step % self.staleness
        staging, ring = self._staging[slot], self._grad_ring[slot]
        # 1. the pipeline launched `staleness` steps ago must be done; adopt its parameters
        if self._wait_slot(slot):
            self.param_flat.copy_(staging)
        # 2. stage this step's gradients (the hooks packed most buckets behind backward already), then run the
        #    pipeline behind the next `staleness` forward/backward passes
        with self._on_comm_stream():
            for b in range(nb):
                if not self._launched[b]:
                    self._launched[b] = True
                    self._pack(b, ring)
        with self._on_comm_stream():
            first = True
            pushed = []
            for b in range(nb):
                pb = self._push(b, ring)
                pushed.append(pb)
                if self.gnorm_sq is None:
                    self._apply(b, pb, staging)
                    self._pull(b, staging)
                elif pb is not None:
                    bk = self.layout.buckets[b]
                    reduce_ops.sumsq(pb[0], self.gnorm_sq, accumulate=not first,
                                     n=bk.shard if self._dist else bk.size, nsrc=pb[1], src_stride=pb[2])
                    first = False
            if self.gnorm_sq is not None:
                if first:
                    self.gnorm_sq.zero_()
                self.comm.all_reduce(self.gnorm_sq)
                for b in range(nb):
                    self._apply(b, pushed[b], staging)
                    self._pull(b, staging)
            self.hyper[2:3].add_(1.0)
            if self.comm_stream is not None:
                ev = torch.cuda.Event()
                ev.record(self.comm_stream)
                self._inflight[slot] = ev
            else:
                self._inflight[slot] = True
        self._last_slot = slot

    def _finish_step(self):
        if self._unpack_needed:
            self._unpack_params()
        if self.comm_stream is not None and (self._prefetch or self.update_mode == "stale"):
            # The packs read p.grad on the comm stream, and in these two modes this stream does NOT wait for the comm
            # stream before it goes on: the caching allocator would hand a freed gradient's block (the BatchNorm
            # gradients live in the compute stream's pool) to the next forward's first allocations while the tail
            # bucket's pack is still queued.  Measured: ResNet-50, live 1-rank group, loss after 48 steps 4.3307 with the
            # prefetch against 4.6683 with the global join and without a communicator (scripts/gpu_r3_ak.sh).  The
            # tensors are kept until this stream HAS waited for those packs: the next forward waits for every pull
            # event (sync), the slot wait `staleness` steps later (stale).
            self._grad_keep.append([p.grad for p in self.params])
            del self._grad_keep[:-max(1, self.staleness if self.update_mode == "stale" else 1)]
        else:
            self._grad_keep.clear()
        for p in self.params:
            p.grad = None
        self._pending = [len(b.params) for b in self.layout.buckets]
        self._launched = [False] * len(self.layout.buckets)
        self.global_step += 1

    def reset_step_state(self, global_step=None):
        """Forget a half-issued step (a hipGraph capture that failed midway ran no kernel but did move the host-side
        bookkeeping): hook counters, launched flags, pending pushes, gradients; optionally the step counter."""
        nb = len(self.layout.buckets)
        for p in self.params:
            p.grad = None
        self._grad_keep.clear()
        self._pending = [len(b.params) for b in self.layout.buckets]
        self._launched = [False] * nb
        self._pushed = [None] * nb
        self._pull_pending = [False] * nb
        if global_step is not None:
            self.global_step = int(global_step)

    def flush(self):
        """Drain a stale-mode pipeline (end of training / before checkpoint): the newest pipeline's output
        holds every update."""
        self._join()
        if self.update_mode == "stale" and self._last_slot is not None:
            self.param_flat.copy_(self._staging[self._last_slot])
            self._inflight = [None] * self.staleness
            self._last_slot = None
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
            # on the comm stream: the shard applies read the value there, and with the pull prefetch or the stale
            # pipeline the previous step's applies may still be queued when the schedule sets the next step's rate --
            # a fill on the compute stream raced with them (ResNet-50 with the bench's warm-up schedule: the tail
            # bucket's apply of step t used the rate of step t + 1, loss after 48 steps 4.3307 against 4.6683)
            with self._on_comm_stream():
                self.hyper[0:1].fill_(self._lr)

    @property
    def lr(self):
        return self._lr

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._attached is not None:
            for h in self._attached["handles"]:
                h.remove()
            self._attached = None
            self._prefetch = False
            self._join()

    def full_master(self):
        """All-gather the fp32 master weights (tests / export): flat [total]."""
        L = self.layout
        self.flush()
        out = torch.zeros(L.total, dtype=torch.float32, device=self.device)
        for bk in L.buckets:
            if self.family == "reduce" and self._dist:
                for j in range(self.num_shards):
                    lo, hi = L.shard_range(bk.index, j)
                    if j == self.comm.rank:
                        out[lo:hi].copy_(self.master[bk.shard_off:bk.shard_off + bk.shard])
                    self.comm.broadcast(out[lo:hi], j)
            else:
                sl = self.master[bk.shard_off:bk.shard_off + bk.shard].contiguous()
                self.comm.all_gather(out[bk.start:bk.start + bk.size], sl)
        return out

    def param_from_flat(self, flat, index):
        s = self.layout.slots[index]
        p = self.params[index]
        return torch.as_strided(flat, p.shape, p.stride(), s.flat_off)
