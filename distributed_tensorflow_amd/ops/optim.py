"""Fused shard-apply optimizers (kernels K1 / K2).

They replace the PS-side ``ApplyMomentum`` / ``ApplyAdam`` ops of the TF1
runtime the reference delegates to (external X4): one pass over the rank's
shard reads the reduced gradient (optionally summing ``nsrc`` received slabs),
the fp32 master weights and the optimizer slots, and writes master, slots and
the wire-dtype copy of the parameters that the pull (all-gather) sends.

``*_reference`` are the plain PyTorch fp32 definitions: the CPU path and the
oracle of ``tests/test_kernels_gpu.py``.
"""
from __future__ import annotations

import torch

from . import hip


def _gsum(g, n, nsrc, src_stride):
    if nsrc <= 1:
        return g[:n].float()
    acc = g[:n].float().clone()
    for s in range(1, nsrc):
        acc += g[s * src_stride:s * src_stride + n].float()
    return acc


def _clip(gnorm_sq, max_norm):
    if gnorm_sq is None or max_norm <= 0:
        return 1.0
    nrm = gnorm_sq.float().reshape(-1)[0].sqrt()
    return torch.where(nrm > max_norm, max_norm / (nrm + 1e-6), torch.ones_like(nrm))


def sgd_momentum_reference(g, p32, m, p_out, *, lr, momentum, wd=0.0, nesterov=False, grad_scale=1.0,
                           nsrc=1, src_stride=0, hyper=None, gnorm_sq=None, max_norm=0.0):
    n = p32.numel()
    if hyper is not None:
        lr = hyper[0]
        grad_scale = grad_scale * hyper[1]
    gs = grad_scale * _clip(gnorm_sq, max_norm)
    gk = _gsum(g, n, nsrc, src_stride) * gs + wd * p32
    m.mul_(momentum).add_(gk)
    upd = gk + momentum * m if nesterov else m
    p32.sub_(lr * upd)
    if p_out is not None:
        p_out[:n].copy_(p32)


def sgd_momentum_apply(g, p32, m, p_out, *, lr, momentum, wd=0.0, nesterov=False, grad_scale=1.0,
                       nsrc=1, src_stride=0, hyper=None, gnorm_sq=None, max_norm=0.0):
    """In place on ``p32``/``m``; writes ``p_out`` (wire dtype) when given."""
    if not hip.on_gpu(g, p32):
        return sgd_momentum_reference(g, p32, m, p_out, lr=lr, momentum=momentum, wd=wd, nesterov=nesterov,
                                      grad_scale=grad_scale, nsrc=nsrc, src_stride=src_stride, hyper=hyper,
                                      gnorm_sq=gnorm_sq, max_norm=max_norm)
    n = p32.numel()
    assert p32.dtype == torch.float32 and m.dtype == torch.float32 and m.numel() == n
    assert g.numel() >= (nsrc - 1) * src_stride + n, "gradient operand too small"
    assert p_out is None or p_out.numel() >= n
    assert g.is_contiguous() and p32.is_contiguous() and m.is_contiguous()
    rc = hip.lib().dtf_sgd_momentum_apply(
        hip.ptr(g), hip.dt(g), nsrc, src_stride, hip.ptr(p32), hip.ptr(m), hip.ptr(p_out),
        hip.dt(p_out) if p_out is not None else 0, n, hip.ptr(hyper), hip.ptr(gnorm_sq),
        float(lr), float(momentum), float(wd), float(grad_scale), float(max_norm), int(bool(nesterov)),
        hip.stream())
    hip.check(rc, "dtf_sgd_momentum_apply")


def adam_reference(g, p32, m, v, p_out, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, adamw=True,
                   bias_correction=True, step=1, grad_scale=1.0, nsrc=1, src_stride=0, hyper=None,
                   gnorm_sq=None, max_norm=0.0):
    n = p32.numel()
    if hyper is not None:
        lr = hyper[0]
        grad_scale = grad_scale * hyper[1]
        step = hyper[2]
    gs = grad_scale * _clip(gnorm_sq, max_norm)
    l2 = 0.0 if adamw else wd
    gk = _gsum(g, n, nsrc, src_stride) * gs + l2 * p32
    m.mul_(beta1).add_((1 - beta1) * gk)
    v.mul_(beta2).add_((1 - beta2) * gk * gk)
    if bias_correction:
        bc1 = 1 - beta1 ** step
        bc2s = (1 - beta2 ** step) ** 0.5
    else:
        bc1, bc2s = 1.0, 1.0
    denom = v.sqrt() / bc2s + eps
    if adamw:
        p32.sub_(lr * wd * p32)
    p32.sub_((lr / bc1) * (m / denom))
    if p_out is not None:
        p_out[:n].copy_(p32)


def adam_apply(g, p32, m, v, p_out, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, adamw=True,
               bias_correction=True, step=1, grad_scale=1.0, nsrc=1, src_stride=0, hyper=None,
               gnorm_sq=None, max_norm=0.0):
    if not hip.on_gpu(g, p32):
        return adam_reference(g, p32, m, v, p_out, lr=lr, beta1=beta1, beta2=beta2, eps=eps, wd=wd,
                              adamw=adamw, bias_correction=bias_correction, step=step, grad_scale=grad_scale,
                              nsrc=nsrc, src_stride=src_stride, hyper=hyper, gnorm_sq=gnorm_sq,
                              max_norm=max_norm)
    n = p32.numel()
    assert p32.dtype == m.dtype == v.dtype == torch.float32 and m.numel() == n and v.numel() == n
    assert g.numel() >= (nsrc - 1) * src_stride + n, "gradient operand too small"
    assert p_out is None or p_out.numel() >= n
    rc = hip.lib().dtf_adam_apply(
        hip.ptr(g), hip.dt(g), nsrc, src_stride, hip.ptr(p32), hip.ptr(m), hip.ptr(v), hip.ptr(p_out),
        hip.dt(p_out) if p_out is not None else 0, n, hip.ptr(hyper), hip.ptr(gnorm_sq), float(lr),
        float(beta1), float(beta2), float(eps), float(wd), float(grad_scale), float(max_norm), float(step),
        int(bool(adamw)), int(bool(bias_correction)), hip.stream())
    hip.check(rc, "dtf_adam_apply")
