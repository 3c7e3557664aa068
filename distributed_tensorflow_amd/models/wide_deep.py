"""Wide&Deep CTR model with large row-sharded embedding tables (config #3; the
embedding scatter-add path of the north star).

* categorical features -> ONE fused, row-sharded table space
  (:class:`~distributed_tensorflow_amd.parallel.embedding.ShardedEmbedding`): each row holds the
  ``embedding_dim`` deep vector plus (padded to 8) the scalar *wide* weight of that id, so
  a step needs one id exchange, one K3 gather and one K4 sparse apply for wide and deep
  together;
* deep tower: concat(embeddings, dense features) -> 1024 -> 512 -> 256 -> 1 (dense
  variables, PS-sharded through the ordinary push/pull buckets);
* wide part: sum of the ids' wide weights + a linear layer on the dense features.

TF names (``tf.estimator.DNNLinearCombinedClassifier`` conventions):
``dnn/hiddenlayer_0/kernel``, ``dnn/logits/bias``, ``linear/linear_model/dense/weights``,
``embeddings/part_<k>`` for the partitioned table.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import embedding as emb_ops
from ..ops.linear import fused_linear
from ..parallel.comm import Comm
from ..parallel.embedding import ShardedEmbedding

NEEDS_COMM = True
WIDE_PAD = 8


class _SplitDeepWide(torch.autograd.Function):
    """``e [B, F, D + pad]`` (+ the dense features) -> ``(h [B, F*D + n_dense + kpad], wide [B, 1] fp32)``: the deep
    tower's input row -- embeddings, dense features, zero columns up to a whole number of K tiles -- written once, and
    the summed wide weights.  Autograd's version of the two slices + reshape + concat copies the largest activation of
    the model three times forward and materialises two zero-filled ``e``-sized gradients backward; here the forward is
    one strided copy into the final buffer and the gradient of ``e`` one concatenation."""

    @staticmethod
    def forward(ctx, e, dense, D, kpad):
        ctx.shape, ctx.D, ctx.dtype = e.shape, D, e.dtype
        # K20 (ops/csrc/rowsplit_kernels.hip) on the GPU: e is read once, h and the wide sums written once
        return emb_ops.split_rows(e, dense, D, dense.shape[1] + kpad)

    @staticmethod
    def backward(ctx, d_h, d_wide):
        return emb_ops.merge_rows(d_h, d_wide, ctx.shape, ctx.D, ctx.dtype), None, None, None


class WideDeep(nn.Module):
    def __init__(self, comm=None, device=None, num_features=26, rows_per_table=100000, embedding_dim=64,
                 dense_features=13, hidden=(1024, 512, 256), emb_optimizer="adagrad", emb_lr=0.05, atomic=False,
                 partition="mod", route="auto", capacity_factor=1.25):
        super().__init__()
        comm = comm or Comm.single()
        if isinstance(rows_per_table, (list, tuple)):
            rows = [int(r) for r in rows_per_table]
        else:
            rows = [int(rows_per_table)] * int(num_features)
        self.num_features, self.embedding_dim = len(rows), int(embedding_dim)
        self.dense_features = int(dense_features)
        self.embeddings = ShardedEmbedding(rows, self.embedding_dim + WIDE_PAD, comm, device, optimizer=emb_optimizer,
                                           lr=emb_lr, atomic=atomic, partition=partition, route=route,
                                           capacity_factor=capacity_factor,
                                           names=["C%d" % (i + 1) for i in range(len(rows))])
        self.embeddings.grad_scale = 1.0 / comm.world_size
        dims = [self.num_features * self.embedding_dim + self.dense_features] + list(hidden)
        self.hidden = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))
        self.logits = nn.Linear(dims[-1], 1)
        self.wide_dense = nn.Linear(self.dense_features, 1)

    def forward(self, ids, dense):
        e = self.embeddings(ids)                                   # [B, F, D + 8]
        if e.is_cuda and e.dtype == torch.bfloat16:
            # K6 / K6T path: bias + ReLU in the GEMM epilogue, dZ and the bias gradient in one pass.  The first layer's
            # 26*64 + 13 = 1677 inputs are padded with zero columns to a whole number of 32-wide K tiles (the parameter
            # keeps its TF shape; the pad of the 3 MB weight is a few microseconds).
            pad = (-self.hidden[0].in_features) % 32
            h, wide = _SplitDeepWide.apply(e, dense.to(e.dtype), self.embedding_dim, pad)
            for i, lin in enumerate(self.hidden):
                w = F.pad(lin.weight, (0, pad)) if (i == 0 and pad) else lin.weight
                h = fused_linear(h, w, lin.bias, act="relu")
            return self.logits(h).float() + wide + self.wide_dense(dense.to(e.dtype)).float()
        h, wide = _SplitDeepWide.apply(e, dense.to(e.dtype), self.embedding_dim, 0)
        for lin in self.hidden:
            h = F.relu(lin(h))
        return self.logits(h).float() + wide + self.wide_dense(dense.to(e.dtype)).float()


OPTIONS = ("num_features", "rows_per_table", "embedding_dim", "dense_features", "emb_optimizer", "emb_lr", "atomic",
           "partition", "route", "capacity_factor")


def build(comm=None, device=None, **kw):
    allowed = OPTIONS
    args = {}
    for k in allowed:
        if k in kw:
            v = kw[k]
            if k in ("num_features", "rows_per_table", "embedding_dim", "dense_features"):
                v = int(v)
            elif k in ("emb_lr", "capacity_factor"):
                v = float(v)
            elif k == "atomic":
                v = str(v).lower() in ("1", "true")
            args[k] = v
    return WideDeep(comm, device, **args)


def prepare(model, device):
    device = torch.device(device)
    emb = model.embeddings
    # the tables are sized for HBM (100+ GB): they were created on their device in their final dtype, so they are
    # detached while the dense tower is moved / cast -- a blanket ``.to(bf16)`` would materialise a second copy
    held = {k: emb._buffers[k] for k in ("table", "accumulator")}
    for k in held:
        emb._buffers[k] = None
    anchor = emb._anchor.data
    model = model.to(device)
    if device.type == "cuda":
        model = model.to(torch.bfloat16)      # dense tower in bf16; tables and accumulators stay fp32
        emb.table_offsets = emb.table_offsets.to(torch.int64)
        emb._anchor.data = anchor.to(device).float()
    for k, v in held.items():
        emb._buffers[k] = None if v is None else v.to(device)
    return model


def tf_variable_names(model):
    out = {}
    for n, _ in model.named_parameters():
        p = n.split(".")
        if p[0] == "hidden":
            out[n] = "dnn/hiddenlayer_%s/%s" % (p[1], "kernel" if p[2] == "weight" else "bias")
        elif p[0] == "logits":
            out[n] = "dnn/logits/" + ("kernel" if p[1] == "weight" else "bias")
        elif p[0] == "wide_dense":
            out[n] = "linear/linear_model/dense/" + ("weights" if p[1] == "weight" else "bias_weights")
    return out


def synthetic_batch(batch, device="cpu", seed=None, model=None, num_features=26, rows_per_table=100000,
                    dense_features=13, **_):
    """Zipf-skewed ids (duplicates inside a batch, like real CTR logs) + a label that
    depends on a few ids so the loss can fall."""
    device = torch.device(device)
    if model is not None:
        rows = model.embeddings.table_rows
        dense_features = model.dense_features
    else:
        rows = [int(rows_per_table)] * int(num_features)
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    u = torch.rand(batch, len(rows), generator=g)
    r = torch.tensor(rows, dtype=torch.float32).unsqueeze(0)
    ids = (r * u.pow(3.0)).long().clamp_(min=0)            # cubic skew: small ids are hot
    ids = torch.minimum(ids, (r - 1).long())
    dense = torch.randn(batch, dense_features, generator=g)
    y = ((ids[:, 0] % 2 == 0) ^ (dense[:, 0] > 0)).float()
    return ids.to(device), dense.to(device), y.to(device)


def loss_fn(model, batch):
    ids, dense, y = batch
    return F.binary_cross_entropy_with_logits(model(ids, dense).squeeze(1), y)


def default_optimizer():
    from ..parallel.ps_runtime import OptimizerSpec

    return OptimizerSpec(kind="adam", lr=1e-3, adamw=False, weight_decay=0.0)
