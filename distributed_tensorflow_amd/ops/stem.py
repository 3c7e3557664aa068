"""The ResNet stem as one autograd node (kernel K15, ``csrc/stem_kernels.hip``):

    max_pool2d(relu(batch_norm(conv2d(x, w, stride=2, padding=3))), k, s, p)        x: [N, 3, H, W], w: [64, 3, 7, 7]

* forward: K15 direct convolution whose epilogue already holds the BatchNorm statistics -> K7 finalize -> K9 pooling
  kernel that applies ``relu(y * scale + shift)`` to every tap on the way in (neither the normalised tensor nor a
  statistics pass over the 411 MB convolution output exists);
* backward: K9 gather -> K7 backward reduction + finalize -> K15 weight gradient, which applies the BatchNorm backward
  (``dx = k1 dy' + k2 y + k3``, ReLU mask recomputed from ``y``) to its operand as it loads it: the gradient w.r.t. the
  convolution output is never written (the input image needs no gradient).

Measured at batch 256, 224 x 224 (``profiles/r2_stem_k15.txt``).  Anything the kernels do not cover (other filter
shapes, dtypes, layouts, CPU, evaluation mode) runs ``ops.pool.bn_relu_max_pool2d(conv(x), bn, ...)``.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import hip

_ENABLED = os.environ.get("DTF_STEM_KERNEL", "1") != "0"         # A/B switch
_WRW = os.environ.get("DTF_STEM_WRW", "1") != "0"                # A/B switch: 0 = BatchNorm apply pass + library weight gradient
# 1: the pool gather also reduces for the BatchNorm backward.  Measured (batch 256): the fused kernel takes 242 us against
# 120 (gather) + 169 (K7 reduction) standalone, but the ResNet-50 step gets SLOWER with it (12 485 -> 12 275 images/s: the
# grid-strided gather streams worse than the flat passes next to the other kernels).  Off by default.
_POOL_RED = os.environ.get("DTF_STEM_POOL_REDUCE", "0") != "0"
_declared = False


def _lib():
    global _declared
    lib = hip.lib()
    if not _declared:
        P, I, L, Fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        hip.declare("dtf_stem_ok", [I, I, I], I)
        hip.declare("dtf_stem_fwd_blocks", [I, I, I], L)
        hip.declare("dtf_stem_pack_w", [P, P, P])
        hip.declare("dtf_stem_fwd", [P, P, P, P, P, I, I, I, P])
        hip.declare("dtf_bn_finalize_pre", [P, P, I, P, P, Fl, Fl, P, P, P, P, P, I, L, I, P])
        hip.declare("dtf_bn_bwd_finalize_pre", [P, I, P, I, P, P, P, P, P, L, I, P])
        hip.declare("dtf_maxpool_bwd_bnred_blocks", [I, I, I, I], L)
        hip.declare("dtf_maxpool_bwd_bnred", [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P])
        if hasattr(lib, "dtf_stem_wrw"):
            hip.declare("dtf_stem_wrw_ok", [I, I, I], I)
            hip.declare("dtf_stem_wrw_workspace", [I, I, I], L)
            hip.declare("dtf_stem_wrw", [P, P, P, P, P, P, P, I, P, I, I, I, P])
        _declared = True
    return lib


_zeros = {}


def _zero_page(dev):
    z = _zeros.get(dev)
    if z is None:
        z = _zeros[dev] = torch.zeros(128, dtype=torch.bfloat16, device=dev)
    return z


def stem_conv_fwd(x, weight):
    """-> ``(conv2d(x, weight, stride=2, padding=3), partials)``; partials: fp32 ``[blocks, 2, 64]`` column sums / sums
    of squares of the (bf16) output.  Caller checks :func:`eligible`."""
    lib = _lib()
    N, _, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dev = x.device
    wp = torch.empty(7 * 64 * 32, dtype=torch.bfloat16, device=dev)
    hip.check(lib.dtf_stem_pack_w(hip.ptr(weight), hip.ptr(wp), hip.stream()), "dtf_stem_pack_w")
    y = torch.empty((N, 64, OH, OW), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    part = torch.empty(int(lib.dtf_stem_fwd_blocks(N, H, W)), 2, 64, dtype=torch.float32, device=dev)
    hip.check(lib.dtf_stem_fwd(hip.ptr(x), hip.ptr(wp), hip.ptr(y), hip.ptr(part), hip.ptr(_zero_page(dev)), N, H, W, hip.stream()),
              "dtf_stem_fwd")
    return y, part


def eligible(x, weight):
    if not (_ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    if tuple(weight.shape) != (64, 3, 7, 7) or x.shape[1] != 3:
        return False
    if not (x.is_contiguous(memory_format=torch.channels_last) and weight.is_contiguous(memory_format=torch.channels_last)):
        return False
    N, _, H, W = x.shape
    return N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1) < (1 << 31) // 64 and bool(_lib().dtf_stem_ok(N, H, W))


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, gamma, beta, running_mean, running_var, momentum, eps, k, stride, pad):
        from . import batchnorm as bnmod
        from . import pool as poolmod

        lib = _lib()
        plib = poolmod._lib()
        dev = x.device
        from . import conv as convmod

        convmod._rt("fwd + bwd", w, x, (2, 2), "K15 stem (7x7 / 2 direct + BN statistics, BN + ReLU inside the pool, "
                                              "BN backward inside the weight gradient)")
        c, part = stem_conv_fwd(x, w)
        N, C, H, W = c.shape
        M = N * H * W
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)       # mean, invstd, scale, shift
        rc = lib.dtf_bn_finalize_pre(hip.ptr(gamma), hip.ptr(beta), hip.dt(gamma), hip.ptr(running_mean), hip.ptr(running_var),
                                     float(momentum), float(eps), stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                                     stats[3].data_ptr(), hip.ptr(part), part.shape[0], M, C, hip.stream())
        hip.check(rc, "dtf_bn_finalize_pre")
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty((N, C, OH, OW), dtype=c.dtype, device=dev, memory_format=torch.channels_last)
        tap = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=dev)
        rc = plib.dtf_maxpool_fwd_affine(hip.ptr(c), stats[2].data_ptr(), stats[3].data_ptr(), hip.ptr(y), hip.ptr(tap), N, H, W,
                                         C, OH, OW, k, stride, pad, hip.stream())
        hip.check(rc, "dtf_maxpool_fwd_affine")
        ctx.save_for_backward(x, w, c, tap, gamma, stats)
        ctx.geom = (N, C, H, W, OH, OW, k, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import batchnorm as bnmod
        from . import pool as poolmod

        x, w, c, tap, gamma, stats = ctx.saved_tensors
        N, C, H, W, OH, OW, k, stride, pad = ctx.geom
        dev = c.device
        lib = _lib()
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        M = N * H * W
        dbn = torch.empty((N, C, H, W), dtype=dy.dtype, device=dev, memory_format=torch.channels_last)
        dgb = torch.empty(2, C, dtype=gamma.dtype, device=dev)
        coef = torch.empty(3, C, dtype=torch.float32, device=dev)
        fused = bool(_WRW and hasattr(lib, "dtf_stem_wrw") and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
                     and lib.dtf_stem_wrw_ok(N, x.shape[2], x.shape[3]))
        dc = None if fused else torch.empty((N, C, H, W), dtype=dy.dtype, device=dev, memory_format=torch.channels_last)
        if fused and _POOL_RED and k == 3 and stride == 2 and 256 % (C // 8) == 0:
            # the pool gather also reduces for the BatchNorm backward (no separate pass over its 411 MB output and z)
            part = torch.empty(int(lib.dtf_maxpool_bwd_bnred_blocks(N, H, W, C)), 2, C, dtype=torch.float32, device=dev)
            rc = lib.dtf_maxpool_bwd_bnred(hip.ptr(dy), hip.ptr(tap), hip.ptr(c), stats[0].data_ptr(), stats[2].data_ptr(),
                                           stats[3].data_ptr(), hip.ptr(dbn), hip.ptr(part), N, H, W, C, OH, OW, k, stride, pad,
                                           hip.stream())
            hip.check(rc, "dtf_maxpool_bwd_bnred")
            rc = lib.dtf_bn_bwd_finalize_pre(hip.ptr(part), part.shape[0], hip.ptr(gamma), hip.dt(gamma), stats[0].data_ptr(),
                                             stats[1].data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), hip.ptr(coef), M, C,
                                             hip.stream())
            hip.check(rc, "dtf_bn_bwd_finalize_pre")
        else:
            rc = poolmod._lib().dtf_maxpool_bwd(hip.ptr(dy), hip.ptr(tap), hip.ptr(dbn), N, H, W, C, OH, OW, k, stride, pad,
                                                hip.stream())
            hip.check(rc, "dtf_maxpool_bwd")
            rc = bnmod._lib().dtf_bn_bwd_train(
                hip.ptr(dbn), hip.ptr(c), None, hip.ptr(gamma), hip.dt(gamma), stats[0].data_ptr(), stats[1].data_ptr(),
                stats[2].data_ptr(), stats[3].data_ptr(), hip.ptr(dc), None, dgb[0].data_ptr(),
                dgb[1].data_ptr(), hip.ptr(coef), hip.ptr(bnmod._partial(dev, M, C)), M, C, 1, 0, 0, 0,
                hip.ptr(bnmod._tickets(dev)), hip.stream())
            hip.check(rc, "dtf_bn_bwd_train (stem)")
        dw = None
        if ctx.needs_input_grad[1]:
            if fused:
                IH, IW = x.shape[2], x.shape[3]
                ws = torch.empty(int(lib.dtf_stem_wrw_workspace(N, IH, IW)), dtype=torch.float32, device=dev)
                dw = torch.empty_like(w)          # channels-last like the parameter: [64][7][7][3] in memory
                rc = lib.dtf_stem_wrw(hip.ptr(dbn), hip.ptr(c), hip.ptr(x), stats[2].data_ptr(), stats[3].data_ptr(),
                                      hip.ptr(coef), hip.ptr(dw), hip.dt(dw), hip.ptr(ws), N, IH, IW, hip.stream())
                hip.check(rc, "dtf_stem_wrw")
            else:
                _, dw, _ = torch.ops.aten.convolution_backward(dc, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                                               [False, True, False])
        dx = None
        if ctx.needs_input_grad[0]:          # an image that wants a gradient: not the training path, library kernel
            dx, _, _ = torch.ops.aten.convolution_backward(dc, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                                           [True, False, False])
        return dx, dw, dgb[0], dgb[1], None, None, None, None, None, None, None


def stem_conv_bn_relu_pool(x, conv, bn, kernel_size, stride=None, padding=0):
    """``max_pool2d(bn(conv(x)), ...)`` for the 7x7 / 2 stem convolution ``conv`` (``nn.Conv2d(3, 64, 7, 2, 3, bias=False)``)
    and a :class:`~.batchnorm.FusedBatchNormAct` with ``relu=True``."""
    from . import batchnorm as bnmod
    from .pool import bn_relu_max_pool2d

    stride = kernel_size if stride is None else stride
    if (bn.training and bn.relu and torch.is_grad_enabled() and conv.bias is None and tuple(conv.stride) == (2, 2)
            and tuple(conv.padding) == (3, 3) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and not x.requires_grad and eligible(x, conv.weight) and kernel_size in (2, 3) and 0 <= padding < kernel_size
            and bn.weight.dtype in (torch.bfloat16, torch.float32) and bn.weight.is_cuda and bn.running_mean is not None
            and bnmod is not None):
        return _StemFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                             int(kernel_size), int(stride), int(padding))
    return bn_relu_max_pool2d(conv(x), bn, kernel_size, stride, padding)
