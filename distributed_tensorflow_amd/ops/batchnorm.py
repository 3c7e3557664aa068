"""Fused BatchNorm (+ residual add) (+ ReLU) conv epilogue (kernel K7).

``FusedBatchNormAct`` is the layer the conv nets of ``models/`` put after every
convolution.  On an MI355X with NHWC bf16 activations it runs the hand-written
kernels of ``csrc/bn_kernels.hip`` (forward: statistics + one fused apply pass;
backward: one reduce pass + one fused apply pass).  Elsewhere (CPU, fp32,
NCHW) it is the plain PyTorch composition ``relu(batch_norm(x) + residual)``,
which is also the oracle for the kernel tests.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip

_declared = False
_CONCAT = __import__("os").environ.get("DTF_BN_CONCAT", "1") != "0"      # A/B switch for bn_act_concat


def _lib():
    global _declared
    lib = hip.lib()
    if not _declared:
        P, I, L, Fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        hip.declare("dtf_bn_partial_floats", [L, I], L)
        hip.declare("dtf_bn_mask_bytes", [L, I], L)
        hip.declare("dtf_bn_fwd_train", [P, P, P, P, P, P, I, P, P, Fl, Fl, P, P, P, P, P, L, I, I, L, L, P])
        hip.declare("dtf_bn_fwd_train_pre", [P, P, P, P, P, P, I, P, P, Fl, Fl, P, P, P, P, P, I, L, I, I, L, P])
        hip.declare("dtf_bn2_fwd_train", [P, P, P, P, P, P, P, P, I, P, P, P, P, Fl, Fl, Fl, Fl, P, P, P, L, I, P])
        hip.declare("dtf_bn2_bwd_train", [P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, L, I, P])
        if hasattr(lib, "dtf_bn2_fwd_train_pre"):
            hip.declare("dtf_bn2_fwd_train_pre",
                        [P, P, P, P, P, P, P, P, I, P, P, P, P, Fl, Fl, Fl, Fl, P, P, P, L, I, P, I, P, I, P])
        hip.declare("dtf_bn_fwd_apply", [P, P, P, P, P, P, L, I, I, P])
        hip.declare("dtf_bn_bwd_train", [P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, L, I, I, L, L, L, P, P])
        _declared = True
    return lib


_partial_ws = {}


_partial_need = {}


def _partial(device, M, C, stream=None):
    """Per-stream scratch for the reduction partials.  ``stream``: the raw handle of the current stream when the
    caller has it already (host time matters here: 100-300 BatchNorm calls per step, a deep ResNet's step is bound by
    host enqueue time -- the size query is cached, the stream looked up once per call)."""
    need = _partial_need.get((M, C))
    if need is None:
        need = _partial_need[(M, C)] = int(_lib().dtf_bn_partial_floats(M, C))
    key = (device, torch.cuda.current_stream(device).cuda_stream if stream is None else stream)
    ws = _partial_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 16), dtype=torch.float32, device=device)
        _partial_ws[key] = ws
    return ws


_mask_need = {}


def _mask_bytes(M, C):
    n = _mask_need.get((M, C))
    if n is None:
        n = _mask_need[(M, C)] = int(_lib().dtf_bn_mask_bytes(M, C))
    return n


_ticket_ws = {}


def _tickets(device, stream=None):
    """Zeroed ticket counters of the calling stream for the backward reduce kernels' fused finalize (``BnFin`` in
    csrc/bn_kernels.hip): the kernels leave them zero again, kernels of one stream run in order, and a second stream
    gets counters of its own."""
    key = (device, torch.cuda.current_stream(device).cuda_stream if stream is None else stream)
    t = _ticket_ws.get(key)
    if t is None:
        t = _ticket_ws[key] = torch.zeros(64, dtype=torch.int32, device=device)
    return t


def _nhwc_rows(x):
    """``x`` -> (M, C) if it is a dense NHWC (channels_last) 4-D tensor or a
    contiguous 2-D [M, C] tensor; else None."""
    if x.dim() == 4:
        N, C, H, W = x.shape
        if x.is_contiguous(memory_format=torch.channels_last):
            return N * H * W, C
        return None
    if x.dim() == 2 and x.is_contiguous():
        return x.shape[0], x.shape[1]
    return None


def _row_stride(x):
    """Row stride (elements) of an NHWC activation the kernels can read in place: ``C`` for a dense tensor, the
    parent's channel count for a channel slice ``parent[:, a:b]`` of a dense NHWC tensor; None otherwise."""
    if _nhwc_rows(x) is not None:
        return x.shape[1]
    if x.dim() == 4:
        N, C, H, W = x.shape
        sn, sc, sh, sw = x.stride()
        if sc == 1 and sw >= C and sw % 8 == 0 and sh == W * sw and (sn == H * W * sw or N == 1) and x.data_ptr() % 16 == 0:
            return sw
    return None


class GradSlices:
    """Shared gradient buffer of a tensor whose channel slices feed several K7 BatchNorm layers (the fused branch
    heads of an Inception block).  The producer of the slices attaches ``(provider, offset, channels)`` to each
    slice as ``_dtf_gradslot``; the BatchNorm backward then writes its ``dx`` straight into that slice of ONE
    buffer (row stride = total channels) and the producer's backward returns the buffer -- no concatenation pass."""

    def __init__(self, shape, dtype, device):
        self.shape, self.dtype, self.device = tuple(shape), dtype, device
        self.buffer = None

    def get(self):
        if self.buffer is None:
            self.buffer = torch.empty(self.shape, dtype=self.dtype, device=self.device, memory_format=torch.channels_last)
        return self.buffer

    def slice(self, off, c):
        return self.get()[:, off:off + c]

    def holds(self, g, off, c):
        b = self.buffer
        return (b is not None and g is not None and g.data_ptr() == b.data_ptr() + off * b.element_size()
                and g.shape[1] == c and g.stride() == b.stride())

    def release(self):
        self.buffer = None


def _dx_target(x, like):
    """-> (dx tensor, row stride) for the gradient of BatchNorm input ``x``."""
    slot = getattr(x, "_dtf_gradslot", None)
    if slot is not None:
        prov, off, c = slot
        if prov.shape[1] % 8 == 0 and off % 8 == 0 and c == like.shape[1] and prov.dtype == like.dtype:
            return prov.slice(off, c), prov.shape[1]
    return torch.empty(like.shape, dtype=like.dtype, device=like.device, memory_format=torch.channels_last) \
        if like.dim() == 4 else torch.empty_like(like), like.shape[1]


def kernel_eligible(x, weight):
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype in (torch.bfloat16, torch.float32)
            and _row_stride(x) is not None and x.shape[1] % 8 == 0 and x.shape[1] <= 5000 and x.numel() > 0)


class _FusedBNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, momentum, eps, relu, pre=None):
        # pre: fp32 [nblk, 2, C] column sums / sums of squares of x from its producer (K6 STATS epilogue,
        # ops/conv.py conv1x1_stats) -- the statistics pass over x is skipped
        lib = _lib()
        ldx = _row_stride(x)             # > C: x is a channel slice of a wider NHWC tensor, read in place
        C = x.shape[1]
        M = x.numel() // C
        dev = x.device
        y = torch.empty(x.shape, dtype=x.dtype, device=dev, memory_format=torch.channels_last) if x.dim() == 4 \
            else torch.empty_like(x)
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
        mask = None
        if residual is not None:
            assert residual.shape == x.shape and residual.dtype == x.dtype and _nhwc_rows(residual) is not None
            if relu:
                mask = torch.empty(_mask_bytes(M, C), dtype=torch.uint8, device=dev)
        sp, st = stats.data_ptr(), hip.stream()          # rows of `stats` by address: no four view tensors per call
        if pre is not None:
            assert ldx == C
            assert pre.dtype == torch.float32 and pre.dim() == 3 and pre.shape[1] == 2 and pre.shape[2] == C and pre.is_contiguous()
            rc = lib.dtf_bn_fwd_train_pre(
                hip.ptr(x), hip.ptr(residual), hip.ptr(y), hip.ptr(mask), hip.ptr(weight), hip.ptr(bias), hip.dt(weight),
                hip.ptr(running_mean), hip.ptr(running_var), float(momentum), float(eps),
                sp, sp + 4 * C, sp + 8 * C, sp + 12 * C,
                hip.ptr(pre), int(pre.shape[0]), M, C, int(relu), 0, st)
        else:
            rc = lib.dtf_bn_fwd_train(
                hip.ptr(x), hip.ptr(residual), hip.ptr(y), hip.ptr(mask), hip.ptr(weight), hip.ptr(bias), hip.dt(weight),
                hip.ptr(running_mean), hip.ptr(running_var), float(momentum), float(eps),
                sp, sp + 4 * C, sp + 8 * C, sp + 12 * C,
                hip.ptr(_partial(dev, M, C, st)), M, C, int(relu), 0, ldx, st)
        hip.check(rc, "dtf_bn_fwd_train")
        ctx.save_for_backward(x, mask, weight, stats)
        # ReLU mask source for backward: 1 = recompute from x (no residual), 3 = bit mask
        ctx.relu_mode = 0 if not relu else (3 if residual is not None else 1)
        ctx.has_res, ctx.dims, ctx.ldx = residual is not None, (M, C), ldx
        ctx.gradslot = getattr(x, "_dtf_gradslot", None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask, weight, stats = ctx.saved_tensors
        M, C = ctx.dims
        lib = _lib()
        dev = x.device
        if _nhwc_rows(dy) is None:
            dy = dy.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else dy.contiguous()
        if ctx.gradslot is not None:
            x._dtf_gradslot = ctx.gradslot
        dx, lddx = _dx_target(x, dy)
        dres = torch.empty_like(dy) if ctx.has_res else None
        dgb = torch.empty(2, C, dtype=weight.dtype, device=dev)
        coef = torch.empty(3, C, dtype=torch.float32, device=dev)
        sp, st = stats.data_ptr(), hip.stream()
        dg, db = dgb[0], dgb[1]
        rc = lib.dtf_bn_bwd_train(
            hip.ptr(dy), hip.ptr(x), hip.ptr(mask), hip.ptr(weight), hip.dt(weight), sp, sp + 4 * C, sp + 8 * C,
            sp + 12 * C, hip.ptr(dx), hip.ptr(dres),
            dg.data_ptr(), db.data_ptr(), hip.ptr(coef), hip.ptr(_partial(dev, M, C, st)), M, C,
            int(ctx.relu_mode), 0, ctx.ldx, lddx, hip.ptr(_tickets(dev, st)), st)
        hip.check(rc, "dtf_bn_bwd_train")
        return dx, dg, db, dres, None, None, None, None, None, None


_BN2 = __import__("os").environ.get("DTF_BN2", "1") != "0"      # A/B switch for fused_bn2_add_relu


class _FusedBN2AddReLU(torch.autograd.Function):
    """``relu(BN_a(xa) + BN_b(xb))`` in one node (first block of a ResNet stage: expand conv + shortcut conv): no
    intermediate shortcut tensor forward, no masked-gradient tensor backward -- 6 activation-sized HBM passes fewer
    than ``bn_a(xa, residual=bn_b(xb))`` (kernels ``bn2_*`` in ``bn_kernels.hip``)."""

    @staticmethod
    def forward(ctx, xa, wa, ba, rma, rva, xb, wb, bb, rmb, rvb, mom_a, eps_a, mom_b, eps_b, pre_a=None, pre_b=None):
        lib = _lib()
        M, C = _nhwc_rows(xa)
        dev = xa.device
        y = torch.empty_like(xa)
        stats = torch.empty(2, 4, C, dtype=torch.float32, device=dev)
        mask = torch.empty(int(lib.dtf_bn_mask_bytes(M, C)), dtype=torch.uint8, device=dev)
        part = _partial2(dev, M, C)
        if pre_a is not None or pre_b is not None:
            # statistics partials [blocks][2][C] from the producing kernels' epilogues (K17): their passes are skipped
            rc = lib.dtf_bn2_fwd_train_pre(
                hip.ptr(xa), hip.ptr(xb), hip.ptr(y), hip.ptr(mask), hip.ptr(wa), hip.ptr(ba), hip.ptr(wb), hip.ptr(bb),
                hip.dt(wa), hip.ptr(rma), hip.ptr(rva), hip.ptr(rmb), hip.ptr(rvb), float(mom_a), float(eps_a), float(mom_b),
                float(eps_b), stats[0].data_ptr(), stats[1].data_ptr(), hip.ptr(part), M, C,
                hip.ptr(pre_a) if pre_a is not None else None, pre_a.shape[0] if pre_a is not None else 0,
                hip.ptr(pre_b) if pre_b is not None else None, pre_b.shape[0] if pre_b is not None else 0, hip.stream())
            hip.check(rc, "dtf_bn2_fwd_train_pre")
        else:
            rc = lib.dtf_bn2_fwd_train(hip.ptr(xa), hip.ptr(xb), hip.ptr(y), hip.ptr(mask), hip.ptr(wa), hip.ptr(ba),
                                       hip.ptr(wb), hip.ptr(bb), hip.dt(wa), hip.ptr(rma), hip.ptr(rva), hip.ptr(rmb),
                                       hip.ptr(rvb), float(mom_a), float(eps_a), float(mom_b), float(eps_b),
                                       stats[0].data_ptr(), stats[1].data_ptr(), hip.ptr(part), M, C, hip.stream())
            hip.check(rc, "dtf_bn2_fwd_train")
        ctx.save_for_backward(xa, xb, mask, wa, wb, stats)
        ctx.dims = (M, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, xb, mask, wa, wb, stats = ctx.saved_tensors
        M, C = ctx.dims
        lib = _lib()
        dev = xa.device
        if _nhwc_rows(dy) is None:
            dy = dy.contiguous(memory_format=torch.channels_last)
        dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
        dgb = torch.empty(2, 2, C, dtype=wa.dtype, device=dev)
        coef = torch.empty(2, 3, C, dtype=torch.float32, device=dev)
        rc = lib.dtf_bn2_bwd_train(hip.ptr(dy), hip.ptr(xa), hip.ptr(xb), hip.ptr(mask), hip.ptr(wa), hip.ptr(wb), hip.dt(wa),
                                   stats[0].data_ptr(), stats[1].data_ptr(), hip.ptr(dxa), hip.ptr(dxb), dgb[0].data_ptr(),
                                   dgb[1].data_ptr(), hip.ptr(coef), hip.ptr(_partial2(dev, M, C)), M, C, hip.stream())
        hip.check(rc, "dtf_bn2_bwd_train")
        return dxa, dgb[0, 0], dgb[0, 1], None, None, dxb, dgb[1, 0], dgb[1, 1], None, None, None, None, None, None, None, None


_partial2_ws = {}


def _partial2(device, M, C):
    need = 2 * int(_lib().dtf_bn_partial_floats(M, C))
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _partial2_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(max(need, 1 << 17), dtype=torch.float32, device=device)
        _partial2_ws[key] = ws
    return ws


def fused_bn2_add_relu(xa, bn_a, xb, bn_b, stats_a=None, stats_b=None):
    """``relu(bn_a(xa) + bn_b(xb))`` for two :class:`FusedBatchNormAct` layers, ``bn_a`` being the one that owns the
    add + ReLU (``relu=True``) and ``bn_b`` the plain one (``relu=False``).  ``stats_a`` / ``stats_b``: statistics
    partials of ``xa`` / ``xb`` from their producers (``conv1x1_stats``), as for ``FusedBatchNormAct(x, stats=...)``."""
    ok = (_BN2 and bn_a.training and bn_b.training and bn_a.relu and not bn_b.relu and torch.is_grad_enabled()
          and xa.shape == xb.shape and xa.dim() == 4 and xa.dtype == xb.dtype
          and kernel_eligible(xa, bn_a.weight) and kernel_eligible(xb, bn_b.weight)
          and _nhwc_rows(xa) is not None and _nhwc_rows(xb) is not None and bn_a.weight.dtype == bn_b.weight.dtype)
    if not ok:
        return bn_a(xa, bn_b(xb, stats=stats_b), stats=stats_a)
    if not hasattr(_lib(), "dtf_bn2_fwd_train_pre"):
        stats_a = stats_b = None
    return _FusedBN2AddReLU.apply(xa, bn_a.weight, bn_a.bias, bn_a.running_mean, bn_a.running_var,
                                  xb, bn_b.weight, bn_b.bias, bn_b.running_mean, bn_b.running_var,
                                  bn_a.momentum, bn_a.eps, bn_b.momentum, bn_b.eps, stats_a, stats_b)


class _BNConcat(torch.autograd.Function):
    """``cat([relu(bn_i(x_i)) ...] + passthrough, dim=1)`` for NHWC tensors in one autograd node: every
    branch's K7 apply pass writes straight into its channel slice of the concat buffer (row stride =
    total channels) and the backward kernels read their slice of the concat gradient in place -- no
    concat copy forward, no slice copies backward (1.0 ms of a 23.5 ms Inception-v3 step).
    args: nb, relu, then per branch (x, weight, bias, running_mean, running_var, momentum, eps), then the
    pass-through tensors (e.g. a pooling branch), which are copied into / out of their slices."""

    @staticmethod
    def forward(ctx, nb, relu, *args):
        lib = _lib()
        branches = [args[7 * i:7 * i + 7] for i in range(nb)]
        extra = list(args[7 * nb:])
        x0 = branches[0][0]
        N, _, H, W = x0.shape
        dev = x0.device
        chans = [b[0].shape[1] for b in branches] + [e.shape[1] for e in extra]
        Ct = sum(chans)
        out = torch.empty((N, Ct, H, W), dtype=x0.dtype, device=dev, memory_format=torch.channels_last)
        M = N * H * W
        saved, off, slots = [], 0, []
        for x, weight, bias, rm, rv, momentum, eps in branches:
            C = x.shape[1]
            stats = torch.empty(4, C, dtype=torch.float32, device=dev)
            rc = lib.dtf_bn_fwd_train(
                hip.ptr(x), None, out.data_ptr() + off * 2, None, hip.ptr(weight), hip.ptr(bias), hip.dt(weight),
                hip.ptr(rm), hip.ptr(rv), float(momentum), float(eps),
                stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
                hip.ptr(_partial(dev, M, C)), M, C, int(relu), Ct, _row_stride(x), hip.stream())
            hip.check(rc, "dtf_bn_fwd_train (concat)")
            saved += [x, weight, stats]
            slots.append(getattr(x, "_dtf_gradslot", None))
            off += C
        for e in extra:
            out.narrow(1, off, e.shape[1]).copy_(e)
            off += e.shape[1]
        ctx.save_for_backward(*saved)
        ctx.nb, ctx.relu, ctx.chans, ctx.M, ctx.slots = nb, relu, chans, M, slots
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib()
        nb, chans, M = ctx.nb, ctx.chans, ctx.M
        if not dout.is_contiguous(memory_format=torch.channels_last):
            dout = dout.contiguous(memory_format=torch.channels_last)
        Ct = dout.shape[1]
        dev = dout.device
        grads, off = [], 0
        for i in range(nb):
            x, weight, stats = ctx.saved_tensors[3 * i:3 * i + 3]
            C = chans[i]
            if ctx.slots[i] is not None:
                x._dtf_gradslot = ctx.slots[i]
            dx, lddx = _dx_target(x, x)
            dgb = torch.empty(2, C, dtype=weight.dtype, device=dev)
            coef = torch.empty(3, C, dtype=torch.float32, device=dev)
            rc = lib.dtf_bn_bwd_train(
                dout.data_ptr() + off * 2, hip.ptr(x), None, hip.ptr(weight), hip.dt(weight), stats[0].data_ptr(),
                stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), hip.ptr(dx), None,
                dgb[0].data_ptr(), dgb[1].data_ptr(), hip.ptr(coef), hip.ptr(_partial(dev, M, C)), M, C,
                1 if ctx.relu else 0, Ct, _row_stride(x), lddx, hip.ptr(_tickets(dev)), hip.stream())
            hip.check(rc, "dtf_bn_bwd_train (concat)")
            grads += [dx, dgb[0], dgb[1], None, None, None, None]
            off += C
        for C in chans[nb:]:
            grads.append(dout.narrow(1, off, C).contiguous(memory_format=torch.channels_last))
            off += C
        return (None, None) + tuple(grads)


def bn_act_concat(pairs, extra=(), relu=True):
    """``torch.cat([bn(x) for x, bn in pairs] + list(extra), 1)`` with ``bn`` a :class:`FusedBatchNormAct`
    (ReLU fused when ``relu``).  K7 + in-place slices for NHWC bf16 training tensors, plain composition
    otherwise."""
    xs = [x for x, _ in pairs]
    ok = _CONCAT and all(kernel_eligible(x, m.weight) and m.training and m.relu == relu and x.dim() == 4 for x, m in pairs) \
        and all(e.is_cuda and e.dtype == xs[0].dtype and e.dim() == 4 and e.shape[1] % 8 == 0 for e in extra) \
        and torch.is_grad_enabled()
    if not ok:
        return torch.cat([m(x) for x, m in pairs] + list(extra), 1)
    args = []
    for x, m in pairs:
        args += [x, m.weight, m.bias, m.running_mean, m.running_var, m.momentum, m.eps]
    extra = [e if e.is_contiguous(memory_format=torch.channels_last) else e.contiguous(memory_format=torch.channels_last)
             for e in extra]
    return _BNConcat.apply(len(pairs), relu, *args, *extra)


def bn_act_reference(x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu):
    y = F.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def fused_bn_act(x, weight, bias, residual=None, running_mean=None, running_var=None, training=True,
                 momentum=0.1, eps=1e-5, relu=True, stats=None):
    """``[relu]( batch_norm(x) [+ residual] )``.  ``stats``: statistics partials of ``x`` from its producer
    (see ``_FusedBNAct``), used in training mode on the kernel path only."""
    if not kernel_eligible(x, weight):
        if x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and \
                not x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 8 == 0:
            x = x.contiguous(memory_format=torch.channels_last)     # keep the conv nets on the kernel path
            if residual is not None:
                residual = residual.contiguous(memory_format=torch.channels_last)
        else:
            w, b = weight, bias
            if x.dtype != weight.dtype and not x.is_cuda:
                w, b = weight.to(x.dtype), bias.to(x.dtype)
            return bn_act_reference(x, w, b, residual, running_mean, running_var, training, momentum, eps, relu)
    if training:
        return _FusedBNAct.apply(x, weight, bias, residual, running_mean, running_var, momentum, eps, relu, stats)
    # inference: statistics are constants -> a single fused apply pass
    lib = _lib()
    if _nhwc_rows(x) is None:          # a channel-slice view: the apply-only entry point reads dense rows
        x = x.contiguous(memory_format=torch.channels_last)
    M, C = _nhwc_rows(x)
    scale = weight.float() * torch.rsqrt(running_var.float() + eps)
    shift = bias.float() - running_mean.float() * scale
    y = torch.empty_like(x)
    mask = torch.empty(int(lib.dtf_bn_mask_bytes(M, C)), dtype=torch.uint8, device=x.device) \
        if (relu and residual is not None) else None
    rc = lib.dtf_bn_fwd_apply(hip.ptr(x), hip.ptr(residual), hip.ptr(y), hip.ptr(mask), hip.ptr(scale),
                              hip.ptr(shift), M, C, int(relu), hip.stream())
    hip.check(rc, "dtf_bn_fwd_apply")
    return y


class FusedBatchNormAct(nn.Module):
    """BatchNorm2d/1d + optional residual + optional ReLU in one layer.
    Parameter names follow ``nn.BatchNorm2d`` (``weight``=gamma, ``bias``=beta,
    ``running_mean``/``running_var`` kept in fp32 whatever the model dtype)."""

    def __init__(self, num_features, relu=True, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.relu, self.eps, self.momentum = num_features, relu, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features, dtype=torch.float32))
        self.register_buffer("running_var", torch.ones(num_features, dtype=torch.float32))

    def _apply(self, fn, recurse=True):
        # keep the moving statistics in fp32 when the model is cast to bf16
        rm, rv = self.running_mean, self.running_var
        super()._apply(fn, recurse)
        for name, old in (("running_mean", rm), ("running_var", rv)):
            new = getattr(self, name)
            if new.dtype != torch.float32:
                setattr(self, name, new.float())
        return self

    def forward(self, x, residual=None, stats=None):
        rm, rv = self.running_mean, self.running_var
        if not x.is_cuda and x.dtype != torch.float32:
            rm, rv = rm.to(x.dtype), rv.to(x.dtype)
            y = bn_act_reference(x, self.weight.to(x.dtype), self.bias.to(x.dtype), residual, rm, rv,
                                 self.training, self.momentum, self.eps, self.relu)
            if self.training:
                self.running_mean.copy_(rm)
                self.running_var.copy_(rv)
            return y
        return fused_bn_act(x, self.weight, self.bias, residual, rm, rv, self.training, self.momentum,
                            self.eps, self.relu, stats)

    def extra_repr(self):
        return "%d, relu=%s, eps=%g, momentum=%g" % (self.num_features, self.relu, self.eps, self.momentum)
