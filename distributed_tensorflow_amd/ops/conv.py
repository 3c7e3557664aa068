"""1x1 stride-1 convolution on NHWC tensors as plain GEMMs where that is faster.

In NHWC a 1x1 convolution *is* a GEMM with no data movement:
``Y[M, Cout] = X[M, Cin] . W[Cout, Cin]^T`` with ``M = N*H*W``.  Measured on MI355X at batch 256
(``scripts/conv1x1_bench.py``, MIOpen with the shipped tuning DB vs hipBLASLt):

* forward:        MIOpen's implicit-GEMM kernels win except for Cin >= 1024;
* data gradient   ``dX = dY . W``: the plain GEMM wins for Cin >= 128 (35 vs 62 us at 14x14, 256->1024);
* weight gradient ``dW = dY^T . X`` (reduction over M ~ 10^5..10^6): MIOpen's split-K kernels win by
  5-10x -- the library GEMM has no good schedule for that shape.

``Conv1x1`` therefore picks per product; the convolution library and the GEMM library are both
"plain library" calls (the hand-written kernels of this package are the fused ops around them).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

_MM_BWD_DATA_MIN_CIN = 128
_MM_FWD_MIN_CIN = 1024
_TN_WRW_MIN_C = 256          # K6T weight gradient when min(Cin, Cout) >= this (narrower layers waste its 256^2 tile)
_TN_WRW = os.environ.get("DTF_CONV1X1_TN", "1") != "0"
_ENABLED = os.environ.get("DTF_CONV1X1_MM", "1") != "0"      # A/B switch: 0 = everything through MIOpen
_FORK = os.environ.get("DTF_CONV1X1_FORK", "1") != "0"        # A/B switch: 0 = autograd sums the shortcut grad


def _wrw(dy, x, weight):
    """dW of a 1x1 stride-1 convolution: K6T (``dY^T . X`` with the pixels as reduction rows, no zero-fill /
    cast passes around it) for wide layers, MIOpen otherwise."""
    cout, cin = weight.shape[0], weight.shape[1]
    if _TN_WRW and min(cin, cout) >= _TN_WRW_MIN_C:
        from .linear import gemm_tn, gemm_tn_eligible

        dy2, x2 = _rows(dy), _rows(x)
        if gemm_tn_eligible(dy2, x2) and weight.dtype == torch.bfloat16:
            return gemm_tn(dy2, x2).view(cout, cin, 1, 1)
    _, dw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [0, 0], [1, 1], False,
                                                   [0, 0], 1, [False, True, False])
    return dw


def _rows(x):
    """NHWC-dense [N,C,H,W] -> contiguous [N*H*W, C] view (no copy)."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


class _Conv1x1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        cout, cin = weight.shape[0], weight.shape[1]
        if cin >= _MM_FWD_MIN_CIN:
            y2 = torch.matmul(_rows(x), weight.reshape(cout, cin).t())
            N, _, H, W = x.shape
            return y2.view(N, H, W, cout).permute(0, 3, 1, 2)
        return F.conv2d(x, weight)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        need_dx, need_dw = ctx.needs_input_grad
        if need_dx and cin >= _MM_BWD_DATA_MIN_CIN:
            N, _, H, W = x.shape
            dx = torch.matmul(_rows(dy), weight.reshape(cout, cin)).view(N, H, W, cin).permute(0, 3, 1, 2)
            need_dx = False
        if need_dx:
            dx, _, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [0, 0], [1, 1], False,
                                                           [0, 0], 1, [True, False, False])
        if need_dw:
            dw = _wrw(dy, x, weight)
        return dx, dw


class _Conv1x1ForkFn(torch.autograd.Function):
    """``(conv1x1(x, w), x)``: the convolution plus a pass-through of its input for a second consumer
    (the residual shortcut).  Autograd would sum the two gradients of ``x`` with a separate elementwise
    kernel (read 2 + write 1 activation-sized tensors, 16 times per ResNet-50 step); here the sum rides
    in the data-gradient GEMM's epilogue instead: ``dX = dY . W + dSkip`` (beta = 1).

    ``own_skip_grad=True`` lets the GEMM accumulate *into* the incoming ``dSkip`` buffer (C == D), which is
    what actually removes the traffic (``scripts/fork_bwd_bench.py``: 1.54 vs 2.81 ms per ResNet-50 step;
    out of place, the library first copies ``dSkip`` into the result and nothing is saved).  Only valid when
    the consumer of the pass-through hands back a gradient tensor nobody else reads -- true for the K7
    residual epilogue and for a convolution's data gradient, false e.g. for ``skip + other`` (autograd gives
    both addends the same tensor)."""

    @staticmethod
    def forward(ctx, x, weight, own_skip_grad):
        ctx.save_for_backward(x, weight)
        ctx.own = bool(own_skip_grad)
        cout, cin = weight.shape[0], weight.shape[1]
        if cin >= _MM_FWD_MIN_CIN:
            N, _, H, W = x.shape
            y = torch.matmul(_rows(x), weight.reshape(cout, cin).t()).view(N, H, W, cout).permute(0, 3, 1, 2)
        else:
            y = F.conv2d(x, weight)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        need_dx, need_dw = ctx.needs_input_grad[:2]
        dx = dw = None
        if dy is None:
            return (dskip if need_dx else None), None, None
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        if need_dx:
            N, _, H, W = x.shape
            w2 = weight.reshape(cout, cin)
            if dskip is None:
                dx2 = torch.matmul(_rows(dy), w2)
            else:
                if not dskip.is_contiguous(memory_format=torch.channels_last):
                    dskip = dskip.contiguous(memory_format=torch.channels_last)
                dx2 = _rows(dskip).addmm_(_rows(dy), w2) if ctx.own else torch.addmm(_rows(dskip), _rows(dy), w2)
            dx = dx2.view(N, H, W, cin).permute(0, 3, 1, 2)
        if need_dw:
            dw = _wrw(dy, x, weight)
        return dx, dw, None


def _eligible(x, weight):
    return (_ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and weight.shape[2] == 1 and weight.shape[3] == 1)


def conv1x1_fork(x, weight, own_skip_grad=False):
    """-> ``(conv1x1(x, weight), x')`` where ``x'`` aliases ``x``; route the *other* consumer of ``x`` (the
    shortcut branch) through ``x'`` and the two gradients are summed inside the data-gradient GEMM.
    See :class:`_Conv1x1ForkFn` for ``own_skip_grad``."""
    if _FORK and _eligible(x, weight) and torch.is_grad_enabled() and x.requires_grad:
        return _Conv1x1ForkFn.apply(x, weight, own_skip_grad)
    return conv1x1(x, weight), x


def conv1x1(x, weight):
    if (_ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and weight.shape[2] == 1 and weight.shape[3] == 1):
        return _Conv1x1Fn.apply(x, weight)
    return F.conv2d(x, weight)


class Conv1x1(nn.Conv2d):
    """``nn.Conv2d(cin, cout, 1, bias=False)`` with the per-product kernel choice above."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, 1, bias=False)

    def forward(self, x):
        return conv1x1(x, self.weight)

    def forward_fork(self, x, own_skip_grad=False):
        return conv1x1_fork(x, self.weight, own_skip_grad)
