"""Row-sharded embedding tables in HBM (SURVEY 2.3 "embedding / table parallel";
replaces TF1's PS-hosted partitioned variables with ``mod`` row partitioning,
external X3, and the ``SparseApply*`` updates on the PS, X4).

All feature tables of a model live in ONE row space (``table_offsets``), sharded
over the W ranks either by ``row % W`` (rank r hosts rows ``r, r+W, r+2W, ...`` as
local rows ``row // W`` -- TF's ``mod`` partition strategy, the default: hot ids of
a skewed distribution spread over all owners) or in contiguous ranges of
``ceil(rows / W)`` (TF's ``div`` strategy); the naming of the checkpoint sections
follows TF's partitioned variables (``<name>/part_<r>``).
Tables are sized for 288 GB of HBM per GPU: a rank's shard (fp32 rows + the
optimizer accumulator) is one contiguous allocation.

One training step does ONE id exchange and ONE row exchange whatever the number
of feature tables::

    forward   ids --all_to_all--> owner: K3 gather rows --all_to_all--> requester
    backward  row grads --all_to_all--> owner: K4 sort/segment sparse apply

The sparse optimizer step is fused into backward on the owning rank (exactly
where TF1 runs SparseApplyAdagrad: on the PS); the table is not a dense
``nn.Parameter`` and never enters the dense push/pull buckets.

Routing (``route=``):

``"exact"``     variable-split exchange: sort by owner, exchange the counts, one host
                read per step.  Never drops a request -- the semantics of the PS-hosted
                partitioned variables this replaces (a TF1 PS answers every ``Gather``).
``"capacity"``  never touches the host: a stable bucket-by-owner HIP kernel
                (``ops/csrc/route_kernels.hip``) places request i at slot
                ``owner * cap + rank_i`` of a fixed-capacity send list, so the three
                all-to-alls have constant, equal splits -- no sort, no ``bincount``, no
                ``.tolist()``; the rows come back through the same slots and are
                un-permuted by a K3 gather.  ``cap = capacity_factor * N / W``
                (MoE-style); requests beyond it are DROPPED (zero rows forward, no
                sparse update) and counted on the device.  Lossy: opt-in only.
``"auto"``      the default.  The first ``calibrate`` lookups run the exact route and
                measure the largest (source, owner) load, all-reduced (MAX) so every
                rank agrees; after that the capacity route runs with ``cap`` sized from
                that measurement (x ``capacity_margin``), i.e. from the data's real skew
                (one low-cardinality feature sends its whole batch to <= 3 owners)
                rather than from a guessed factor.  The overflow counter is polled after
                capacity lookups 1, 2, 4, 8, ... and then every ``overflow_check_every``;
                an overflow grows the capacity and sends the module back to the exact
                route for another calibration.

:meth:`check_overflow` is a COLLECTIVE: the per-rank counters (counted on the sending
rank) are all-reduced (MAX) before anybody decides to grow, so ``capacity_factor`` -- and
with it the split sizes of the equal-split all-to-alls -- stays identical on every rank.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..ops import embedding as emb_ops
from .comm import Comm


class _Lookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, module, flat_ids):
        out, saved = module._forward_impl(flat_ids)
        ctx.module, ctx.saved = module, saved
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ctx.module._backward_impl(grad_out.contiguous(), ctx.saved)
        return None, None, None


class ShardedEmbedding(nn.Module):
    def __init__(self, table_rows, dim, comm: Comm = None, device=None, table_dtype=torch.float32,
                 out_dtype=None, optimizer="adagrad", lr=0.01, eps=1e-10, init_std=0.01, atomic=False,
                 names=None, seed=1234, initial_accumulator=0.1, partition="mod", route="auto",
                 capacity_factor=1.25, overflow_check_every=200, calibrate=2, capacity_margin=1.5):
        super().__init__()
        self.comm = comm or Comm.single()
        self.dim = int(dim)
        self.table_rows = [int(r) for r in table_rows]
        self.names = list(names) if names else ["embedding_%d" % i for i in range(len(self.table_rows))]
        offs, tot = [], 0
        for r in self.table_rows:
            offs.append(tot)
            tot += r
        self.total_rows = tot
        assert partition in ("mod", "div") and route in ("auto", "capacity", "exact")
        self.partition, self.route = partition, route
        self.capacity_factor = float(capacity_factor)
        self.overflow_check_every = int(overflow_check_every)
        self.calibrate, self.capacity_margin = max(1, int(calibrate)), float(capacity_margin)
        self._calibrating = self.calibrate if route == "auto" else 0     # exact lookups left before the switch
        self._max_pair_load = 0.0       # largest (source, owner) load seen while calibrating, relative to n / W
        self._routed_lookups = 0
        W, r = self.comm.world_size, self.comm.rank
        self.rows_per = (tot + W - 1) // W                       # div: rows per part
        self.local_rows = self.part_rows(r, W)
        device = torch.device(device or self.comm.device)
        self.optimizer, self.lr, self.eps, self.atomic = optimizer, float(lr), float(eps), bool(atomic)
        self.out_dtype = out_dtype or (torch.bfloat16 if device.type == "cuda" else torch.float32)
        g = torch.Generator().manual_seed(seed + r)
        # the shard is initialised row by row from a per-rank stream; for a W-independent init use
        # load_rows().  Large shards are filled on the device to avoid a host copy.
        if device.type == "cuda":
            gd = torch.Generator(device=device).manual_seed(seed + r)
            table = torch.empty(self.local_rows, self.dim, dtype=torch.float32, device=device).normal_(0, init_std, generator=gd)
        else:
            table = torch.empty(self.local_rows, self.dim, dtype=torch.float32).normal_(0, init_std, generator=g)
        self.register_buffer("table", table.to(table_dtype))
        if optimizer == "adagrad":
            acc = torch.full((self.local_rows, self.dim), float(initial_accumulator), dtype=torch.float32, device=device)
        elif optimizer == "rowwise_adagrad":
            acc = torch.full((self.local_rows,), float(initial_accumulator), dtype=torch.float32, device=device)
        else:
            acc = None
        self.register_buffer("accumulator", acc)
        self.register_buffer("table_offsets", torch.tensor(offs, dtype=torch.int64, device=device))
        self.register_buffer("overflow", torch.zeros(1, dtype=torch.int64, device=device), persistent=False)
        # a dummy tensor that requires grad keeps autograd calling our backward
        self._anchor = nn.Parameter(torch.zeros(1, device=device), requires_grad=True)
        self._anchor.dtf_skip_ps = True
        self.grad_scale = 1.0
        self._lookups = 0
        self.dropped_total = 0

    # ------------------------------------------------------------ partition
    def part_rows(self, k, W=None, strategy=None, total=None):
        """Rows of the fused row space hosted by part ``k`` of ``W``."""
        W = W or self.comm.world_size
        total = self.total_rows if total is None else total
        if (strategy or self.partition) == "mod":
            return (total - k + W - 1) // W if total > k else 0
        per = (total + W - 1) // W
        return max(0, min(per, total - k * per))

    def global_rows_of(self, k, W=None, strategy=None, lo=0, hi=None, total=None):
        """Global ids of local rows ``[lo, hi)`` of part ``k`` (a 1-D int64 tensor)."""
        W = W or self.comm.world_size
        n = self.part_rows(k, W, strategy, total)
        hi = n if hi is None else min(hi, n)
        i = torch.arange(lo, hi, dtype=torch.int64)
        if (strategy or self.partition) == "mod":
            return k + W * i
        total = self.total_rows if total is None else total
        return k * ((total + W - 1) // W) + i

    def _owner_local(self, g, W=None, strategy=None, total=None):
        W = W or self.comm.world_size
        if (strategy or self.partition) == "mod":
            return g % W, torch.div(g, W, rounding_mode="floor")
        total = self.total_rows if total is None else total
        per = (total + W - 1) // W
        o = torch.div(g, per, rounding_mode="floor")
        return o, g - o * per

    # ------------------------------------------------------------------ API
    def forward(self, ids):
        """``ids``: int64 ``[B, F]`` (feature f indexes table f) -> ``[B, F, dim]``."""
        B, F = ids.shape
        assert F == len(self.table_rows)
        flat = (ids + self.table_offsets.unsqueeze(0)).reshape(-1)
        out = _Lookup.apply(self._anchor, self, flat)
        return out.view(B, F, self.dim)

    def lookup_flat(self, global_rows):
        """Global row ids ``[n]`` -> ``[n, dim]`` (already offset into the fused row space)."""
        return _Lookup.apply(self._anchor, self, global_rows.reshape(-1))

    def capacity(self, n):
        """Slots per (source, owner) pair for ``n`` requests per rank."""
        W = self.comm.world_size
        if W == 1:
            return max(1, n)
        c = int(n / W * self.capacity_factor) + 1
        return max(64, min(n, (c + 63) // 64 * 64))

    def check_overflow(self, grow=True):
        """The only host read of the routed path, and a COLLECTIVE (every rank must call it at the same lookup
        count -- ``_forward_impl`` does): the largest number of requests any rank dropped since the last check.
        The counter lives on the SENDING rank (``route_kernels.hip``), so without the all-reduce only the ranks
        that overflowed would grow their factor and the next equal-split all-to-all would be issued with
        different sizes on different ranks (an RCCL hang, or rows in the wrong slots).  On overflow every rank
        grows the factor by the same x1.5; ``route="auto"`` additionally re-calibrates on the exact route."""
        if self.comm.live and self.comm.world_size > 1:
            worst = self.overflow.clone()
            self.comm.all_reduce(worst, op=torch.distributed.ReduceOp.MAX)
            dropped, mine = int(worst.item()), int(self.overflow.item())
        else:
            dropped = mine = int(self.overflow.item())
        if dropped:
            self.overflow.zero_()
            self.dropped_total += mine
            msg = "[embedding] up to %d lookups per rank (%d here) exceeded the all-to-all capacity (factor %.2f) " \
                  "and were dropped" % (dropped, mine, self.capacity_factor)
            if grow:
                self.capacity_factor *= 1.5
                msg += "; capacity factor -> %.2f on every rank" % self.capacity_factor
                if self.route == "auto":
                    self._calibrating = self.calibrate
                    msg += "; re-calibrating on the exact route"
            print(msg, flush=True)
        return dropped

    def _poll_due(self):
        """Capacity lookups 1, 2, 4, 8, ... (a bad factor shows on the first step, not after 200), then every
        ``overflow_check_every``.  A function of the lookup count alone: identical on every rank."""
        k, every = self._routed_lookups, self.overflow_check_every
        if not every:
            return False
        if k < every:
            return k & (k - 1) == 0
        return k % every == 0

    def _calibrated(self, sc, n):
        """One exact lookup of the calibration phase is done: fold its largest (source, owner) load into the
        running maximum (all-reduced: one agreed number); after the last one, size the capacity from it."""
        W = self.comm.world_size
        load = torch.tensor([max(sc) / max(n / W, 1.0)], dtype=torch.float32, device=self.table.device)
        if self.comm.live and W > 1:
            self.comm.all_reduce(load, op=torch.distributed.ReduceOp.MAX)
        self._max_pair_load = max(self._max_pair_load, float(load.item()))
        self._calibrating -= 1
        if self._calibrating == 0:
            self.capacity_factor = max(self.capacity_factor, self._max_pair_load * self.capacity_margin)
            self._max_pair_load = 0.0
            self._routed_lookups = 0
            if self.comm.rank == 0:
                print("[embedding] capacity factor %.2f from the measured owner load (x%.2f margin)" % (
                    self.capacity_factor, self.capacity_margin), flush=True)

    # ------------------------------------------------------------ internals
    def _route_exact(self, flat_ids):
        """Sort requests by owner rank and exchange the split sizes (one host sync)."""
        W = self.comm.world_size
        dest, local = self._owner_local(flat_ids)
        order = torch.sort(dest, stable=True).indices
        send_ids = local[order]
        send_counts = torch.bincount(dest, minlength=W)
        recv_counts = torch.empty_like(send_counts)
        self.comm.all_to_all(recv_counts, send_counts)
        # one host sync per step: all_to_all_single needs the split sizes on the host
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        return order, send_ids, sc, rc

    def _forward_impl(self, flat_ids):
        W = self.comm.world_size
        if W == 1 and not self.comm.live:
            rows = emb_ops.embedding_gather(self.table, flat_ids, self.out_dtype)
            return rows, ("local", flat_ids)
        dev = flat_ids.device
        if self.route == "exact" or self._calibrating:
            order, send_ids, sc, rc = self._route_exact(flat_ids)
            if self._calibrating:
                self._calibrated(sc, flat_ids.numel())
            local = torch.empty(sum(rc), dtype=torch.int64, device=dev)
            self.comm.all_to_all(local, send_ids, rc, sc)
            rows = emb_ops.embedding_gather(self.table, local, self.out_dtype)
            back = torch.empty(flat_ids.numel(), self.dim, dtype=self.out_dtype, device=dev)
            self.comm.all_to_all(back, rows, sc, rc)
            out = torch.empty_like(back)
            out[order] = back
            return out, ("exact", local, order, sc, rc)
        # fixed-capacity routed exchange: everything below is device work with constant shapes
        n = flat_ids.numel()
        cap = self.capacity(n)
        send_ids, slot, inv, _ = emb_ops.embedding_route(flat_ids.contiguous(), W, cap, self.partition == "div",
                                                         self.rows_per, self.total_rows, self.overflow)
        recv_ids = torch.empty(W * cap, dtype=torch.int64, device=dev)
        self.comm.all_to_all(recv_ids, send_ids)
        rows = emb_ops.embedding_gather(self.table, recv_ids, self.out_dtype)          # empty slots -> zero rows
        back = torch.empty(W * cap, self.dim, dtype=self.out_dtype, device=dev)
        self.comm.all_to_all(back, rows)
        out = emb_ops.embedding_gather(back, slot, self.out_dtype)                     # un-permute (dropped -> zeros)
        self._lookups += 1
        self._routed_lookups += 1
        if self._poll_due():
            self.check_overflow()
        return out, ("routed", recv_ids, inv)

    def _backward_impl(self, grad_out, saved):
        kind = saved[0]
        if kind == "local":
            local = saved[1]
        elif kind == "exact":
            _, local, order, sc, rc = saved
            send = grad_out[order].contiguous()
            recv = torch.empty(sum(rc), self.dim, dtype=grad_out.dtype, device=grad_out.device)
            self.comm.all_to_all(recv, send, rc, sc)
            grad_out = recv
        else:
            _, local, inv = saved
            send = emb_ops.embedding_gather(grad_out, inv, grad_out.dtype)             # slot order, empty -> zeros
            recv = torch.empty_like(send)
            self.comm.all_to_all(recv, send)
            grad_out = recv
        emb_ops.embedding_sparse_apply(self.table, self.accumulator, local, grad_out, self.optimizer, self.lr,
                                       self.eps, self.grad_scale, 0.0, 1, self.atomic)

    # -------------------------------------------------- checkpoint plumbing
    def checkpoint_sections(self):
        """Sections this rank contributes to its checkpoint shard."""
        r = self.comm.rank
        sec = {"embeddings/part_%d" % r: self.table}
        if self.accumulator is not None:
            sec["embeddings/part_%d/Adagrad" % r] = self.accumulator
        return sec

    def checkpoint_variables(self):
        """Index entries (chief): every table is a partitioned variable whose part k is the
        rows of the fused row space held by rank k (``mod``: ``k, k+W, ...``; ``div``: a contiguous range)."""
        W = self.comm.world_size
        out = {}
        for k in range(W):
            rows_k = self.part_rows(k)
            part = {"strategy": self.partition, "num_parts": W, "part": k, "total_rows": self.total_rows,
                    "tables": dict(zip(self.names, self.table_rows))}
            out["embeddings/part_%d" % k] = {
                "dtype": str(self.table.dtype).replace("torch.", ""), "shape": [rows_k, self.dim],
                "partition": part,
                "pieces": [{"rank": k, "section": "embeddings/part_%d" % k, "offset": 0,
                            "length": rows_k * self.dim, "var_offset": 0}]}
            if self.accumulator is not None:
                width = self.dim if self.accumulator.dim() == 2 else 1
                out["embeddings/part_%d/Adagrad" % k] = {
                    "dtype": "float32", "shape": [rows_k, width] if width > 1 else [rows_k], "partition": part,
                    "pieces": [{"rank": k, "section": "embeddings/part_%d/Adagrad" % k, "offset": 0,
                                "length": rows_k * width, "var_offset": 0}]}
        return out

    def restore(self, reader, chunk_rows=1 << 20):
        """Load this rank's rows (table + accumulator) from a checkpoint written with ANY world size and
        partition strategy: local row i is global row g(i); the old part ``owner_old(g)`` holds it at
        ``local_old(g)``.  Streams ``chunk_rows`` rows at a time (tables are sized for HBM, not host RAM)."""
        v0 = reader.index["variables"].get("embeddings/part_0")
        if v0 is None:
            raise KeyError("checkpoint has no sharded embedding sections")
        old = v0["partition"]
        W_old, strat_old, total_old = int(old["num_parts"]), old.get("strategy", "mod"), int(old["total_rows"])
        if total_old != self.total_rows:
            raise ValueError("checkpoint tables have %d rows, the model %d" % (total_old, self.total_rows))
        r, W = self.comm.rank, self.comm.world_size
        targets = [("", self.table, self.dim)]
        if self.accumulator is not None and reader.has("embeddings/part_0/Adagrad"):
            targets.append(("/Adagrad", self.accumulator, self.dim if self.accumulator.dim() == 2 else 1))
        same = W_old == W and strat_old == self.partition
        for lo in range(0, self.local_rows, chunk_rows):
            hi = min(self.local_rows, lo + chunk_rows)
            if same:
                for suffix, dst, width in targets:
                    flat = reader.read_flat("embeddings/part_%d%s" % (r, suffix), lo * width, (hi - lo) * width)
                    dst[lo:hi].copy_(flat.view((hi - lo, width) if dst.dim() == 2 else (hi - lo,)))
                continue
            g = self.global_rows_of(r, lo=lo, hi=hi)
            own, loc = self._owner_local(g, W_old, strat_old, total_old)
            for k in range(W_old):
                sel = (own == k).nonzero(as_tuple=True)[0]
                if sel.numel() == 0:
                    continue
                lk = loc[sel]
                a, b = int(lk.min()), int(lk.max()) + 1
                for suffix, dst, width in targets:
                    span = reader.read_flat("embeddings/part_%d%s" % (k, suffix), a * width, (b - a) * width)
                    span = span.view(b - a, width)[lk - a]
                    dst[lo + sel] = span.to(dst.dtype).to(dst.device) if dst.dim() == 2 else \
                        span[:, 0].to(dst.dtype).to(dst.device)

    def load_rows(self, global_rows, values):
        """Set rows given by GLOBAL ids (only those this rank owns are written)."""
        own, loc = self._owner_local(global_rows)
        mine = own == self.comm.rank
        self.table[loc[mine]] = values[mine].to(self.table.dtype)

    def dense_rows(self, global_rows):
        """Fetch rows by global id from the local shard (debug / tests, W == 1 or owned rows)."""
        return self.table[self._owner_local(global_rows)[1]]
