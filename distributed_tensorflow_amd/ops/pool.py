"""Max pooling for NHWC bf16 activations (kernel K9, ``csrc/pool_kernels.hip``).

The framework pooling kernels cost 0.86 ms of a 28 ms ResNet-50 step on MI355X (forward 250 us,
backward 607 us + a 49 us zero fill) for ~1.5 GB of traffic.  K9 records which tap of each window
won (1 byte/element) and runs the backward as a gather over the windows covering each input pixel:
every ``dx`` element is written once, nothing is zero-filled, ``x`` is not re-read.
Other layouts / dtypes / devices use ``F.max_pool2d`` (also the test oracle).
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn.functional as F

from . import hip

_declared = False
_ENABLED = os.environ.get("DTF_POOL_KERNEL", "1") != "0"     # A/B switch


def _lib():
    global _declared
    lib = hip.lib()
    if not _declared:
        P, I = ctypes.c_void_p, ctypes.c_int
        hip.declare("dtf_maxpool_fwd", [P, P, P, I, I, I, I, I, I, I, I, I, P])
        hip.declare("dtf_maxpool_bwd", [P, P, P, I, I, I, I, I, I, I, I, I, P])
        hip.declare("dtf_avgpool", [P, P, I, I, I, I, I, I, I, I, I, I, I, ctypes.c_int64, P])
        hip.declare("dtf_maxpool_fwd_affine", [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P])
        _declared = True
    return lib


class _MaxPoolNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        N, C, H, W = x.shape
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        tap = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
        rc = _lib().dtf_maxpool_fwd(hip.ptr(x), hip.ptr(y), hip.ptr(tap), N, H, W, C, OH, OW, k, stride, pad,
                                    hip.stream())
        hip.check(rc, "dtf_maxpool_fwd")
        ctx.save_for_backward(tap)
        ctx.geom = (N, C, H, W, OH, OW, k, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        (tap,) = ctx.saved_tensors
        N, C, H, W, OH, OW, k, stride, pad = ctx.geom
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        rc = _lib().dtf_maxpool_bwd(hip.ptr(dy), hip.ptr(tap), hip.ptr(dx), N, H, W, C, OH, OW, k, stride, pad,
                                    hip.stream())
        hip.check(rc, "dtf_maxpool_bwd")
        return dx, None, None, None


def max_pool2d(x, kernel_size, stride=None, padding=0):
    """``F.max_pool2d`` (square window, ceil_mode off, no dilation); K9 for NHWC bf16 CUDA tensors."""
    stride = kernel_size if stride is None else stride
    if (_ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and kernel_size in (2, 3)
            and x.shape[1] % 8 == 0 and x.is_contiguous(memory_format=torch.channels_last)
            and 0 <= padding < kernel_size and x.shape[2] + 2 * padding >= kernel_size
            and x.shape[3] + 2 * padding >= kernel_size):
        return _MaxPoolNHWC.apply(x, int(kernel_size), int(stride), int(padding))
    return F.max_pool2d(x, kernel_size, stride, padding)


_BN_POOL = os.environ.get("DTF_BN_POOL", "1") != "0"      # A/B switch for bn_relu_max_pool2d
# Its backward without the full-size gradient tensor (the K7 passes gather the pool gradient themselves).  Bitwise
# equal but measured SLOWER (ResNet-50 -0.8 %, Inception-v3 -1.7 %): the gather runs twice (reduce + apply) and its
# index arithmetic / dependent loads take the two passes off the HBM roofline.  Off by default.
_BN_POOL_BWD = os.environ.get("DTF_BN_POOL_BWD", "0") != "0"


class _BNReLUMaxPool(torch.autograd.Function):
    """``max_pool2d(relu(batch_norm(x)))`` in training mode without the normalised full-resolution tensor: the K7
    statistics pass + finalize, then the K9 pooling kernel applies ``relu(x * scale + shift)`` to every tap on the way
    in.  Backward: K9 gather -> gradient w.r.t. the (virtual) BatchNorm output, then the ordinary K7 backward, which
    recomputes the ReLU mask from ``x`` anyway."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, k, stride, pad):
        from . import batchnorm as bnmod

        lib = _lib()
        bl = bnmod._lib()
        if not getattr(_BNReLUMaxPool, "_declared", False):
            P, I, L, Fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
            hip.declare("dtf_bn_stats_train", [P, P, P, I, P, P, Fl, Fl, P, P, P, P, P, L, I, P])
            _BNReLUMaxPool._declared = True
        N, C, H, W = x.shape
        M = N * H * W
        dev = x.device
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)
        rc = bl.dtf_bn_stats_train(hip.ptr(x), hip.ptr(weight), hip.ptr(bias), hip.dt(weight), hip.ptr(running_mean),
                                   hip.ptr(running_var), float(momentum), float(eps), stats[0].data_ptr(), stats[1].data_ptr(),
                                   stats[2].data_ptr(), stats[3].data_ptr(), hip.ptr(bnmod._partial(dev, M, C)), M, C,
                                   hip.stream())
        hip.check(rc, "dtf_bn_stats_train")
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        tap = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=dev)
        rc = lib.dtf_maxpool_fwd_affine(hip.ptr(x), stats[2].data_ptr(), stats[3].data_ptr(), hip.ptr(y), hip.ptr(tap), N, H, W,
                                        C, OH, OW, k, stride, pad, hip.stream())
        hip.check(rc, "dtf_maxpool_fwd_affine")
        ctx.save_for_backward(x, tap, weight, stats)
        ctx.geom = (N, C, H, W, OH, OW, k, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import batchnorm as bnmod

        x, tap, weight, stats = ctx.saved_tensors
        N, C, H, W, OH, OW, k, stride, pad = ctx.geom
        dev = x.device
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        M = N * H * W
        dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dev, memory_format=torch.channels_last)
        dgb = torch.empty(2, C, dtype=weight.dtype, device=dev)
        coef = torch.empty(3, C, dtype=torch.float32, device=dev)
        if _BN_POOL_BWD:
            # the K7 backward passes gather the pool's gradient themselves: no full-resolution gradient tensor
            if not getattr(_BNReLUMaxPool, "_declared_bwd", False):
                P, I = ctypes.c_void_p, ctypes.c_int
                hip.declare("dtf_bn_bwd_train_pool", [P, P, P, P, I, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P])
                _BNReLUMaxPool._declared_bwd = True
            rc = bnmod._lib().dtf_bn_bwd_train_pool(
                hip.ptr(dy), hip.ptr(tap), hip.ptr(x), hip.ptr(weight), hip.dt(weight), stats[0].data_ptr(), stats[1].data_ptr(),
                stats[2].data_ptr(), stats[3].data_ptr(), hip.ptr(dx), dgb[0].data_ptr(), dgb[1].data_ptr(), hip.ptr(coef),
                hip.ptr(bnmod._partial(dev, M, C)), N, H, W, C, OH, OW, k, stride, pad, hip.stream())
            hip.check(rc, "dtf_bn_bwd_train_pool")
            return dx, dgb[0], dgb[1], None, None, None, None, None, None, None
        dbn = torch.empty((N, C, H, W), dtype=dy.dtype, device=dev, memory_format=torch.channels_last)
        rc = _lib().dtf_maxpool_bwd(hip.ptr(dy), hip.ptr(tap), hip.ptr(dbn), N, H, W, C, OH, OW, k, stride, pad, hip.stream())
        hip.check(rc, "dtf_maxpool_bwd")
        rc = bnmod._lib().dtf_bn_bwd_train(
            hip.ptr(dbn), hip.ptr(x), None, hip.ptr(weight), hip.dt(weight), stats[0].data_ptr(), stats[1].data_ptr(),
            stats[2].data_ptr(), stats[3].data_ptr(), hip.ptr(dx), None, dgb[0].data_ptr(), dgb[1].data_ptr(), hip.ptr(coef),
            hip.ptr(bnmod._partial(dev, M, C)), M, C, 1, 0, 0, 0, hip.ptr(bnmod._tickets(dev)), hip.stream())
        hip.check(rc, "dtf_bn_bwd_train (pool)")
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, None


def bn_relu_max_pool2d(x, bn, kernel_size, stride=None, padding=0):
    """``max_pool2d(bn(x), ...)`` for a :class:`~.batchnorm.FusedBatchNormAct` with ``relu=True``."""
    from . import batchnorm as bnmod

    stride = kernel_size if stride is None else stride
    if (_BN_POOL and _ENABLED and bn.training and bn.relu and torch.is_grad_enabled() and x.dim() == 4
            and bnmod.kernel_eligible(x, bn.weight) and x.is_contiguous(memory_format=torch.channels_last)
            and kernel_size in (2, 3) and 0 <= padding < kernel_size and x.shape[2] + 2 * padding >= kernel_size
            and x.shape[3] + 2 * padding >= kernel_size):
        return _BNReLUMaxPool.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                    int(kernel_size), int(stride), int(padding))
    return max_pool2d(bn(x), kernel_size, stride, padding)


class _AvgPoolNHWC(torch.autograd.Function):
    """``x`` may be a channel slice of a wider NHWC tensor (read in place through its pixel stride); a slice that
    carries a ``_dtf_gradslot`` (ops/batchnorm.py GradSlices) gets its gradient written into the shared buffer."""

    @staticmethod
    def forward(ctx, x, k, stride, pad, include_pad):
        from . import batchnorm as bnmod

        N, C, H, W = x.shape
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty((N, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        rc = _lib().dtf_avgpool(hip.ptr(x), hip.ptr(y), 0, N, H, W, C, OH, OW, k, stride, pad, int(include_pad),
                                bnmod._row_stride(x), hip.stream())
        hip.check(rc, "dtf_avgpool fwd")
        ctx.geom = (N, C, H, W, OH, OW, k, stride, pad, int(include_pad))
        ctx.gradslot = getattr(x, "_dtf_gradslot", None)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, OH, OW, k, stride, pad, inc = ctx.geom
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx, ld = None, C
        if ctx.gradslot is not None:
            prov, off, c = ctx.gradslot
            if prov.shape[1] % 8 == 0 and off % 8 == 0 and c == C and prov.dtype == dy.dtype:
                dx, ld = prov.slice(off, c), prov.shape[1]
        if dx is None:
            dx = torch.empty((N, C, H, W), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last)
        rc = _lib().dtf_avgpool(hip.ptr(dy), hip.ptr(dx), 1, N, H, W, C, OH, OW, k, stride, pad, inc, ld, hip.stream())
        hip.check(rc, "dtf_avgpool bwd")
        return dx, None, None, None, None


def _pixel_stride_ok(x):
    from . import batchnorm as bnmod

    return bnmod._row_stride(x) is not None


def avg_pool2d(x, kernel_size, stride=None, padding=0, count_include_pad=True):
    """``F.avg_pool2d`` (square window, ceil_mode off); K9 for NHWC bf16 CUDA tensors.  The framework's
    own NHWC kernels take 220 us forward / 400 us backward per 35x35x288 call at batch 128 (19 % of an
    Inception-v3 step); these run at the HBM rate (~35 us each)."""
    stride = kernel_size if stride is None else stride
    if (_ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and kernel_size in (2, 3, 5)
            and x.shape[1] % 8 == 0 and _pixel_stride_ok(x)
            and 0 <= padding < kernel_size and x.shape[2] + 2 * padding >= kernel_size
            and x.shape[3] + 2 * padding >= kernel_size):
        return _AvgPoolNHWC.apply(x, int(kernel_size), int(stride), int(padding), bool(count_include_pad))
    return F.avg_pool2d(x, kernel_size, stride, padding, count_include_pad=count_include_pad)


class _GlobalAvgPoolNHWC(torch.autograd.Function):
    """``x.mean((2, 3))`` whose gradient is written directly as a dense NHWC tensor.  Autograd's own backward hands the
    consumer (a K7 BatchNorm backward) a broadcast view; making that channels_last went through the framework's
    generic strided copy: 85 us for ResNet-50's [256, 2048, 7, 7] (51 MB), every step."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return x.mean((2, 3))

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = ctx.shape
        dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
        dx.copy_((dy * (1.0 / (H * W))).view(N, 1, 1, C).expand(N, H, W, C))      # contiguous rows, broadcast source
        return dx.permute(0, 3, 1, 2)


def global_avg_pool(x):
    """``x.mean((2, 3))`` for an NHWC-dense 4-D tensor (any other input: the framework op)."""
    if x.dim() == 4 and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last) and x.requires_grad:
        return _GlobalAvgPoolNHWC.apply(x)
    return x.mean((2, 3))
