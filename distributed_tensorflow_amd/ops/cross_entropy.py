"""Softmax cross-entropy on bf16 logits (kernel K12, ``csrc/ce_kernels.hip``).

``F.cross_entropy(logits.float(), labels)`` -- the usual way to keep the loss in fp32 -- materialises the fp32
logits, their log-softmax and both gradients: for BERT's MLM head ([4864, 30522] per 64-sequence step) that is
~1.3 ms of pure HBM traffic.  K12 reads the bf16 row once per direction and does the arithmetic in fp32
registers.  Mean reduction over the rows whose label is not ``ignore_index`` (the semantics of
``F.cross_entropy``); everything the kernel does not cover falls back to it.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn.functional as F

from . import hip

_declared = False
_ENABLED = os.environ.get("DTF_CE_KERNEL", "1") != "0"       # A/B switch


def _lib():
    global _declared
    lib = hip.lib()
    if not _declared:
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        hip.declare("dtf_ce_fwd", [P, P, P, P, L, I, L, P])
        hip.declare("dtf_ce_bwd", [P, P, P, P, P, L, I, L, P])
        _declared = True
    return lib


def kernel_eligible(logits, labels):
    return (_ENABLED and logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2 and logits.is_contiguous()
            and labels.dtype == torch.int64 and labels.dim() == 1 and labels.shape[0] == logits.shape[0]
            and logits.shape[0] > 0 and logits.data_ptr() % 16 == 0)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        M, V = logits.shape
        labels = labels.contiguous()
        out = torch.empty(2, M, dtype=torch.float32, device=logits.device)      # lse, per-row loss
        rc = _lib().dtf_ce_fwd(hip.ptr(logits), hip.ptr(labels), out[0].data_ptr(), out[1].data_ptr(), M, V, int(ignore_index),
                               hip.stream())
        hip.check(rc, "dtf_ce_fwd")
        valid = (labels != ignore_index)
        inv = 1.0 / valid.sum().clamp(min=1).float()
        ctx.save_for_backward(logits, labels, out, valid, inv)
        ctx.ignore_index = int(ignore_index)
        return out[1].sum() * inv

    @staticmethod
    def backward(ctx, g):
        logits, labels, out, valid, inv = ctx.saved_tensors
        M, V = logits.shape
        row_scale = (valid.float() * (inv * g.float())).contiguous()
        dl = torch.empty_like(logits)
        rc = _lib().dtf_ce_bwd(hip.ptr(logits), hip.ptr(labels), out[0].data_ptr(), hip.ptr(row_scale), hip.ptr(dl), M, V,
                               ctx.ignore_index, hip.stream())
        hip.check(rc, "dtf_ce_bwd")
        return dl, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    """Mean softmax cross-entropy, fp32 result; ``logits [M, V]``, ``labels [M]``."""
    if kernel_eligible(logits, labels):
        return _CrossEntropy.apply(logits, labels, ignore_index)
    return F.cross_entropy(logits.float(), labels, ignore_index=ignore_index)
