"""Row-sharded embedding tables in HBM (SURVEY 2.3 "embedding / table parallel";
replaces TF1's PS-hosted partitioned variables with ``mod`` row partitioning,
external X3, and the ``SparseApply*`` updates on the PS, X4).

All feature tables of a model live in ONE row space (``table_offsets``), sharded
by ``row % W`` over the W ranks: rank r hosts rows ``r, r+W, r+2W, ...`` as local
rows ``row // W`` -- TF's ``mod`` partition strategy, and the naming of the
checkpoint sections follows TF's partitioned variables (``<name>/part_<r>``).
Tables are sized for 288 GB of HBM per GPU: a rank's shard (fp32 rows + the
optimizer accumulator) is one contiguous allocation.

One training step does ONE id exchange and ONE row exchange whatever the number
of feature tables::

    forward   ids --all_to_all--> owner: K3 gather rows --all_to_all--> requester
    backward  row grads --all_to_all--> owner: K4 sort/segment sparse apply

The sparse optimizer step is fused into backward on the owning rank (exactly
where TF1 runs SparseApplyAdagrad: on the PS); the table is not a dense
``nn.Parameter`` and never enters the dense push/pull buckets.
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
                 names=None, seed=1234, initial_accumulator=0.1):
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
        W, r = self.comm.world_size, self.comm.rank
        self.local_rows = (tot - r + W - 1) // W if tot > r else 0
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
        # a dummy tensor that requires grad keeps autograd calling our backward
        self._anchor = nn.Parameter(torch.zeros(1, device=device), requires_grad=True)
        self._anchor.dtf_skip_ps = True
        self.grad_scale = 1.0

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

    # ------------------------------------------------------------ internals
    def _route(self, flat_ids):
        """Sort requests by owner rank and exchange the split sizes."""
        W = self.comm.world_size
        dest = flat_ids % W
        order = torch.sort(dest, stable=True).indices
        send_ids = flat_ids[order]
        send_counts = torch.bincount(dest, minlength=W)
        recv_counts = torch.empty_like(send_counts)
        self.comm.all_to_all(recv_counts, send_counts)
        # one host sync per step: all_to_all_single needs the split sizes on the host
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        return order, send_ids, sc, rc

    def _forward_impl(self, flat_ids):
        W = self.comm.world_size
        if W == 1:
            rows = emb_ops.embedding_gather(self.table, flat_ids, self.out_dtype)
            return rows, (flat_ids, None, None, None)
        order, send_ids, sc, rc = self._route(flat_ids)
        recv_ids = torch.empty(sum(rc), dtype=torch.int64, device=flat_ids.device)
        self.comm.all_to_all(recv_ids, send_ids, rc, sc)
        local = recv_ids // W
        rows = emb_ops.embedding_gather(self.table, local, self.out_dtype)
        back = torch.empty(flat_ids.numel(), self.dim, dtype=self.out_dtype, device=flat_ids.device)
        self.comm.all_to_all(back, rows, sc, rc)
        out = torch.empty_like(back)
        out[order] = back
        return out, (local, order, sc, rc)

    def _backward_impl(self, grad_out, saved):
        local, order, sc, rc = saved
        W = self.comm.world_size
        if W > 1:
            send = grad_out[order].contiguous()
            recv = torch.empty(sum(rc), self.dim, dtype=grad_out.dtype, device=grad_out.device)
            self.comm.all_to_all(recv, send, rc, sc)
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
        rows ``k, k+W, ...`` of the fused row space held by rank k."""
        W = self.comm.world_size
        out = {}
        for k in range(W):
            rows_k = (self.total_rows - k + W - 1) // W if self.total_rows > k else 0
            out["embeddings/part_%d" % k] = {
                "dtype": str(self.table.dtype).replace("torch.", ""), "shape": [rows_k, self.dim],
                "partition": {"strategy": "mod", "num_parts": W, "part": k, "total_rows": self.total_rows,
                              "tables": dict(zip(self.names, self.table_rows))},
                "pieces": [{"rank": k, "section": "embeddings/part_%d" % k, "offset": 0,
                            "length": rows_k * self.dim, "var_offset": 0}]}
        return out

    def load_rows(self, global_rows, values):
        """Set rows given by GLOBAL ids (only those this rank owns are written)."""
        W, r = self.comm.world_size, self.comm.rank
        mine = (global_rows % W) == r
        self.table[global_rows[mine] // W] = values[mine].to(self.table.dtype)

    def dense_rows(self, global_rows):
        """Fetch rows by global id from the local shard (debug / tests, W == 1 or owned rows)."""
        W = self.comm.world_size
        return self.table[global_rows // W]
