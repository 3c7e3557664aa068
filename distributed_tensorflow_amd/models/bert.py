"""BERT-base pre-training (config #4: PS-sharded Adam, bf16, sequence length 512).

L=12, H=768, A=12, intermediate 3072, vocab 30522, masked-LM + next-sentence heads; ~110 M
parameters whose Adam state (fp32 master + m + v) is sharded over the PS ranks and updated
by the K2 fused Adam shard-apply kernel.  Linear layers are :class:`FusedLinear` (K6 MFMA
GEMM with the bias / GELU / residual epilogue fused, K6T weight gradients), LayerNorm is K10 and
attention is K11 (``ops/attention.py``; PyTorch SDPA when a mask or dropout is requested).

Variables are named like google-research/bert checkpoints: ``bert/embeddings/word_embeddings``,
``bert/encoder/layer_0/attention/self/query/kernel``, ``cls/predictions/transform/dense/kernel``
(TF stores kernels ``[in, out]``; the flat shard bytes here are torch's ``[out, in]``, recorded
as such in the checkpoint index).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops import attention
from ..ops.layernorm import FusedLayerNorm
from ..ops.cross_entropy import cross_entropy
from ..ops.embedding import dense_embedding
from ..ops.linear import FusedLinear, linear_residual_layernorm


class _SplitQKV(torch.autograd.Function):
    """Packed projection output ``[B, S, 3, A, D]`` -> q, k, v views ``[B, A, S, D]`` for the attention
    kernel.  Backward gathers dq, dk, dv straight into the packed layout with one copy kernel; plain
    autograd stacks them as ``[3, B, A, S, D]`` and the next op then re-copies that into token-major
    order (two passes over the 3H-wide tensor per layer: 276 us at batch 64)."""

    @staticmethod
    def forward(ctx, qkv):
        q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
        return q, k, v

    @staticmethod
    def backward(ctx, dq, dk, dv):
        return torch.stack([dq.transpose(1, 2), dk.transpose(1, 2), dv.transpose(1, 2)], dim=2)


class Layer(nn.Module):
    def __init__(self, H, A, I, p, act="gelu_tanh"):
        super().__init__()
        self.A = A
        self.qkv = FusedLinear(H, 3 * H)
        self.attn_out = FusedLinear(H, H)
        self.attn_ln = FusedLayerNorm(H, eps=1e-12)
        self.inter = FusedLinear(H, I, act=act)
        self.out = FusedLinear(I, H)
        self.out_ln = FusedLayerNorm(H, eps=1e-12)
        self.p = p

    def forward(self, x, mask, kv_len=None):
        B, S, H = x.shape
        # x feeds the projection and the residual: the fork sums both gradients inside the dX GEMM
        qkv, x = self.qkv.forward_fork(x, own_skip_grad=True)
        qkv5 = qkv.view(B, S, 3, self.A, H // self.A)
        drop = self.p if self.training else 0.0
        if drop == 0.0 and attention.kernel_eligible(qkv5):
            # K11: consumes the packed projection in place, writes the context token-major, and its backward
            # writes dQ / dK / dV straight into the packed gradient -- no transposes or copies either way.
            # The padding mask is a prefix mask: handed over as one length per sequence.
            ctx = attention.attention_packed(qkv5, kv_len)
        else:
            q, k, v = _SplitQKV.apply(qkv5)
            ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=drop)
            ctx = ctx.transpose(1, 2).reshape(B, S, H)
        x = linear_residual_layernorm(ctx, self.attn_out, x, self.attn_ln)      # residual in the GEMM (beta = 1), bias in K10
        h, x = self.inter.forward_fork(x, own_skip_grad=True)
        return linear_residual_layernorm(h, self.out, x, self.out_ln)


class Bert(nn.Module):
    def __init__(self, vocab=30522, hidden=768, layers=12, heads=12, intermediate=3072, max_pos=512,
                 type_vocab=2, dropout=0.0, act="gelu_tanh"):
        super().__init__()
        self.hidden, self.vocab = hidden, vocab
        self.word_embeddings = nn.Embedding(vocab, hidden)
        self.position_embeddings = nn.Embedding(max_pos, hidden)
        self.token_type_embeddings = nn.Embedding(type_vocab, hidden)
        self.emb_ln = FusedLayerNorm(hidden, eps=1e-12)
        # act: "gelu_tanh" is the google-research/bert definition (tanh form); "gelu" the erf form
        self.layers = nn.ModuleList(Layer(hidden, heads, intermediate, dropout, act) for _ in range(layers))
        self.pooler = FusedLinear(hidden, hidden)
        self.mlm_transform = FusedLinear(hidden, hidden, act=act)
        self.mlm_ln = FusedLayerNorm(hidden, eps=1e-12)
        self.mlm_bias = nn.Parameter(torch.zeros(vocab))
        self.nsp = nn.Linear(hidden, 2)
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, (nn.Linear, nn.Embedding)):
            nn.init.trunc_normal_(m.weight, std=0.02, a=-0.04, b=0.04)
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)

    def forward(self, input_ids, token_type_ids, attention_mask, masked_positions):
        B, S = input_ids.shape
        pos = torch.arange(S, device=input_ids.device)
        x = dense_embedding(input_ids, self.word_embeddings.weight) + self.position_embeddings(pos).unsqueeze(0) \
            + dense_embedding(token_type_ids, self.token_type_embeddings.weight)
        x = self.emb_ln(x)
        mask = kv_len = None
        if attention_mask is not None:
            # BERT's input mask marks real tokens with 1 and trailing padding with 0 (a prefix mask)
            mask = attention_mask[:, None, None, :].to(torch.bool)
            kv_len = attention_mask.sum(-1, dtype=torch.int32)
        for layer in self.layers:
            x = layer(x, mask, kv_len)
        pooled = torch.tanh(self.pooler(x[:, 0]))
        # masked-LM head on the masked positions only (max_predictions_per_seq)
        idx = masked_positions.unsqueeze(-1).expand(-1, -1, self.hidden)
        h = torch.gather(x, 1, idx).reshape(-1, self.hidden)
        h = self.mlm_ln(self.mlm_transform(h))
        mlm_logits = F.linear(h, self.word_embeddings.weight, self.mlm_bias)      # tied decoder
        return mlm_logits, self.nsp(pooled)


def build(layers=12, hidden=768, heads=12, intermediate=3072, vocab=30522, seq_len=512, act="gelu_tanh", **_):
    return Bert(int(vocab), int(hidden), int(layers), int(heads), int(intermediate), max(512, int(seq_len)), act=act)


def prepare(model, device):
    device = torch.device(device)
    if device.type == "cuda":
        return model.to(device=device, dtype=torch.bfloat16)
    return model.to(device)


def tf_variable_names(model):
    """torch name -> TF name, or -> ``[(TF name, row0, row1), ...]`` for the packed QKV projection, which the
    checkpoint index lists as google-research/bert's three variables ``attention/self/{query,key,value}/{kernel,bias}``
    (row ranges of the same shard bytes: output rows are ordered q, k, v)."""
    out = {}
    ln = {"weight": "gamma", "bias": "beta"}
    kb = {"weight": "kernel", "bias": "bias"}
    H = model.hidden
    for n, _ in model.named_parameters():
        p = n.split(".")
        if p[0] == "layers" and p[2] == "qkv":
            base = "bert/encoder/layer_%s/attention/self/" % p[1]
            out[n] = [(base + "%s/%s" % (part, kb[p[3]]), i * H, (i + 1) * H)
                      for i, part in enumerate(("query", "key", "value"))]
            continue
        if p[0] in ("word_embeddings", "position_embeddings", "token_type_embeddings"):
            tf = "bert/embeddings/" + p[0]
        elif p[0] == "emb_ln":
            tf = "bert/embeddings/LayerNorm/" + ln[p[1]]
        elif p[0] == "layers":
            base = "bert/encoder/layer_%s/" % p[1]
            sub = {"attn_out": "attention/output/dense/",
                   "attn_ln": "attention/output/LayerNorm/", "inter": "intermediate/dense/",
                   "out": "output/dense/", "out_ln": "output/LayerNorm/"}[p[2]]
            tf = base + sub + (ln[p[3]] if p[2].endswith("ln") else kb[p[3]])
        elif p[0] == "pooler":
            tf = "bert/pooler/dense/" + kb[p[1]]
        elif p[0] == "mlm_transform":
            tf = "cls/predictions/transform/dense/" + kb[p[1]]
        elif p[0] == "mlm_ln":
            tf = "cls/predictions/transform/LayerNorm/" + ln[p[1]]
        elif p[0] == "mlm_bias":
            tf = "cls/predictions/output_bias"
        elif p[0] == "nsp":
            tf = "cls/seq_relationship/" + ("output_weights" if p[1] == "weight" else "output_bias")
        else:
            tf = n
        out[n] = tf
    return out


def synthetic_batch(batch, device="cpu", seed=None, seq_len=512, vocab=30522, max_predictions=76, **_):
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    seq_len = int(seq_len)
    ids = torch.randint(1000, vocab, (batch, seq_len), generator=g)
    types = (torch.arange(seq_len).unsqueeze(0) >= seq_len // 2).long().expand(batch, -1).contiguous()
    mask = torch.ones(batch, seq_len, dtype=torch.long)
    npred = min(max_predictions, max(1, int(seq_len * 0.15)))
    pos = torch.stack([torch.randperm(seq_len, generator=g)[:npred] for _ in range(batch)])
    labels = torch.gather(ids, 1, pos)
    ids = ids.scatter(1, pos, 103)          # [MASK]
    nsp = torch.randint(0, 2, (batch,), generator=g)
    return tuple(t.to(device) for t in (ids, types, mask, pos, labels, nsp))


def loss_fn(model, batch):
    ids, types, mask, pos, labels, nsp = batch
    # "no padding" is decided once per mask tensor: the check reads a device value (a host sync), which per
    # step would drain the launch queue
    full = getattr(mask, "_dtf_all_ones", None)
    if full is None:
        full = bool((mask == 1).all())
        mask._dtf_all_ones = full
    mlm_logits, nsp_logits = model(ids, types, None if full else mask, pos)
    # K12: the [B*76, 30522] MLM logits stay bf16 in memory, softmax arithmetic in fp32 registers
    return cross_entropy(mlm_logits, labels.reshape(-1)) + F.cross_entropy(nsp_logits.float(), nsp)


def default_optimizer():
    from ..parallel.ps_runtime import OptimizerSpec

    # BERT's AdamWeightDecayOptimizer: decoupled decay 0.01, no bias correction, and its exclusion list
    # ("LayerNorm", "layer_norm", "bias"): those variables form their own bucket group applied with decay 0
    return OptimizerSpec(kind="adam", lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.01, adamw=True,
                         bias_correction=False, max_grad_norm=1.0, no_decay=("_ln.", ".bias", "mlm_bias"))
