"""Self-attention over a packed projection (kernel K11, ``csrc/attn_kernels.hip``).

``attention_packed(qkv)`` takes the fused QKV projection output ``[B, S, 3, A, 64]`` (bf16, CUDA) and
returns the context ``[B, S, A*64]`` -- the layouts on both sides of BERT's attention, so nothing is
transposed or copied around the kernels.  Head dim 64, S in {128, 256, 384, 512}, optional key-padding lengths, no dropout;
anything else goes through ``F.scaled_dot_product_attention`` (also the oracle of the kernel tests).
"""
from __future__ import annotations

import ctypes
import math
import os

import torch
import torch.nn.functional as F

from . import hip

_declared = False
_ENABLED = os.environ.get("DTF_ATTN_KERNEL", "1") != "0"      # A/B switch


def _lib():
    global _declared
    lib = hip.lib()
    if not _declared:
        P, I, L, Fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        hip.declare("dtf_attn_fwd", [P, P, P, P, P, L, L, L, I, I, I, I, I, I, I, Fl, P, P])
        hip.declare("dtf_attn_bwd", [P, P, P, P, P, P, P, P, P, P, L, L, L, I, I, I, L, L, L, I, I, I, I, I, I, I, Fl, P, P])
        _declared = True
    return lib


def kernel_eligible(qkv):
    return (_ENABLED and qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 5 and qkv.shape[2] == 3
            and qkv.shape[4] == 64 and qkv.shape[1] in (128, 256, 384, 512) and qkv.is_contiguous())


def attn_fwd_packed(qkv, kv_len=None):
    """Raw forward: -> (ctx [B, S, A*64], lse [B, A, S] fp32 in the log2 domain)."""
    B, S, _, A, D = qkv.shape
    o = torch.empty(B, S, A * D, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, A, S, dtype=torch.float32, device=qkv.device)
    base, esz = qkv.data_ptr(), 2
    rs, bs = 3 * A * D, S * 3 * A * D
    rc = _lib().dtf_attn_fwd(base, base + A * D * esz, base + 2 * A * D * esz, hip.ptr(o), hip.ptr(lse), bs, bs, bs,
                             rs, rs, rs, B, A, S, D, 1.0 / math.sqrt(D), hip.ptr(kv_len), hip.stream())
    hip.check(rc, "dtf_attn_fwd")
    return o, lse


def attn_bwd_packed(qkv, o, lse, do, kv_len=None):
    """Raw backward: -> dqkv [B, S, 3, A, 64] (every element written by the kernels)."""
    B, S, _, A, D = qkv.shape
    if not do.is_contiguous():
        do = do.contiguous()
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty(B, A, S, dtype=torch.float32, device=qkv.device)
    base, gbase, esz = qkv.data_ptr(), dqkv.data_ptr(), 2
    rs, bs = 3 * A * D, S * 3 * A * D
    hd = A * D * esz
    rc = _lib().dtf_attn_bwd(base, base + hd, base + 2 * hd, hip.ptr(o), hip.ptr(do), hip.ptr(lse), hip.ptr(dsum),
                             gbase, gbase + hd, gbase + 2 * hd, bs, bs, bs, rs, rs, rs, bs, bs, bs, rs, rs, rs,
                             B, A, S, D, 1.0 / math.sqrt(D), hip.ptr(kv_len), hip.stream())
    hip.check(rc, "dtf_attn_bwd")
    return dqkv


class _AttentionPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, kv_len):
        o, lse = attn_fwd_packed(qkv, kv_len)
        ctx.save_for_backward(qkv, o, lse, kv_len)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, kv_len = ctx.saved_tensors
        return attn_bwd_packed(qkv, o, lse, do, kv_len), None


def attention_packed(qkv, kv_len=None):
    """``softmax(q k^T / sqrt(64)) v`` per head for the packed projection ``[B, S, 3, A, 64]`` -> ``[B, S, A*64]``.
    ``kv_len`` (int32 ``[B]``, optional): keys at positions >= kv_len[b] are padding and get no weight
    (BERT's attention mask is such a prefix mask); padded QUERY rows still produce (meaningless) output."""
    if kernel_eligible(qkv):
        if kv_len is not None:
            kv_len = kv_len.to(device=qkv.device, dtype=torch.int32).contiguous()
        return _AttentionPacked.apply(qkv, kv_len)
    return attention_reference(qkv, kv_len)


def attention_reference(qkv, kv_len=None):
    B, S, _, A, D = qkv.shape
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
    mask = None
    if kv_len is not None:
        mask = (torch.arange(S, device=qkv.device)[None, :] < kv_len.to(qkv.device)[:, None])[:, None, None, :]
    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask).transpose(1, 2).reshape(B, S, A * D)
