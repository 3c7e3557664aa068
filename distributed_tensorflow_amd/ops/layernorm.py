"""LayerNorm over the last dimension (kernel K10, ``csrc/ln_kernels.hip``).

Single-pass forward and backward for bf16 CUDA activations (one wave per row, the row held in
registers; dgamma / dbeta reduced deterministically through per-block partial rows).  Everything
else -- CPU, fp32 activations, H % 8 != 0, H > 4096 -- uses ``F.layer_norm``, which is also the
oracle of the kernel tests.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip

_declared = False
_ws = {}
_ENABLED = os.environ.get("DTF_LN_KERNEL", "1") != "0"       # A/B switch


def _lib():
    global _declared
    lib = hip.lib()
    if not _declared:
        P, I, L, Fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        hip.declare("dtf_ln_partial_floats", [L, I], L)
        hip.declare("dtf_ln_fwd", [P, P, P, P, I, P, P, P, L, I, Fl, P])
        hip.declare("dtf_ln_bwd", [P, P, P, P, P, P, I, P, P, P, P, L, I, P])
        _declared = True
    return lib


def _partial(dev, n):
    w = _ws.get(dev)
    if w is None or w.numel() < n:
        w = torch.empty(max(n, 1 << 18), dtype=torch.float32, device=dev)
        _ws[dev] = w
    return w


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, pre_bias=None):
        H = x.shape[-1]
        x2 = x.reshape(-1, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        y = torch.empty_like(x2)
        stats = torch.empty(2, M, dtype=torch.float32, device=x.device)
        rc = _lib().dtf_ln_fwd(hip.ptr(x2), hip.ptr(pre_bias), hip.ptr(weight), hip.ptr(bias), hip.dt(weight), hip.ptr(y),
                               stats[0].data_ptr(), stats[1].data_ptr(), M, H, float(eps), hip.stream())
        hip.check(rc, "dtf_ln_fwd")
        ctx.save_for_backward(x2, weight, stats, pre_bias)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, stats, pre_bias = ctx.saved_tensors
        M, H = x2.shape
        dy2 = dy.reshape(M, H)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        lib = _lib()
        dx = torch.empty_like(x2)
        dg, db = torch.empty_like(weight), torch.empty_like(weight)
        part = _partial(dy.device, int(lib.dtf_ln_partial_floats(M, H)))
        rc = lib.dtf_ln_bwd(hip.ptr(dy2), hip.ptr(x2), hip.ptr(pre_bias), stats[0].data_ptr(), stats[1].data_ptr(), hip.ptr(weight),
                            hip.dt(weight), hip.ptr(dx), hip.ptr(dg), hip.ptr(db), hip.ptr(part), M, H, hip.stream())
        hip.check(rc, "dtf_ln_bwd")
        # pre_bias gets NO gradient here: it is the deferred bias of the projection in front, whose backward computes
        # that gradient (the column sum of dx) inside its weight-gradient kernel
        return dx.view(ctx.shape), dg, db, None, None


def kernel_eligible(x, weight, bias):
    H = x.shape[-1]
    return (_ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and weight is not None and bias is not None
            and weight.dtype in (torch.bfloat16, torch.float32) and bias.dtype == weight.dtype
            and H % 8 == 0 and H <= 4096 and x.numel() > 0)


def layer_norm(x, weight, bias, eps=1e-5):
    if kernel_eligible(x, weight, bias):
        return _LayerNorm.apply(x, weight, bias, eps)
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def layer_norm_deferred_bias(x, pre_bias, weight, bias, eps=1e-5):
    """``layer_norm(x + pre_bias)`` where the GRADIENT of ``pre_bias`` is produced by the caller (see
    ``ops/linear.py: linear_residual_layernorm``).  Kernel path only -- check :func:`kernel_eligible` first."""
    assert kernel_eligible(x, weight, bias) and pre_bias.dtype == weight.dtype and pre_bias.numel() == x.shape[-1]
    return _LayerNorm.apply(x, weight, bias, eps, pre_bias)


class FusedLayerNorm(nn.LayerNorm):
    """``nn.LayerNorm(H)`` (last dimension, affine) on the K10 kernels."""

    def forward(self, x):
        if len(self.normalized_shape) == 1 and self.elementwise_affine:
            return layer_norm(x, self.weight, self.bias, self.eps)
        return super().forward(x)
