"""Norm / cast helpers (kernel K8): replaces TF
``clip_by_global_norm`` pieces and dtype casts around the PS traffic."""
from __future__ import annotations

import torch

from . import hip

_ws = {}


def _workspace(device):
    w = _ws.get(device)
    if w is None:
        w = torch.empty(2048, dtype=torch.float32, device=device)
        _ws[device] = w
    return w


def sumsq(x, out, accumulate=False, n=None, nsrc=1, src_stride=0):
    """``out[0] (+)= sum(x**2)`` in fp32 with a fixed reduction order.  With ``nsrc > 1`` the operand is
    ``nsrc`` slabs of ``n`` elements ``src_stride`` apart (an all-to-all receive buffer) that are summed
    first: the norm of the reduced gradient, without a separate reduction pass."""
    x = x.reshape(-1)
    n = x.numel() if n is None else int(n)
    if not x.is_cuda:
        acc = x[:n].float()
        for s in range(1, nsrc):
            acc = acc + x[s * src_stride:s * src_stride + n].float()
        s = (acc ** 2).sum()
        if accumulate:
            out[0] += s
        else:
            out[0] = s
        return out
    assert out.dtype == torch.float32 and x.is_contiguous()
    assert x.numel() >= (nsrc - 1) * src_stride + n, "sumsq operand too small"
    rc = hip.lib().dtf_sumsq(hip.ptr(x), hip.dt(x), n, int(nsrc), int(src_stride), hip.ptr(_workspace(x.device)),
                             hip.ptr(out), int(accumulate), hip.stream())
    hip.check(rc, "dtf_sumsq")
    return out


def cast_scale(src, dst, scale=1.0):
    """``dst = cast(src * scale)`` for flat fp32/bf16 buffers."""
    assert src.numel() == dst.numel()
    if not src.is_cuda:
        dst.copy_((src.float() * scale))
        return dst
    assert src.is_contiguous() and dst.is_contiguous()
    rc = hip.lib().dtf_cast_scale(hip.ptr(src), hip.dt(src), hip.ptr(dst), hip.dt(dst), src.numel(),
                                  float(scale), hip.stream())
    hip.check(rc, "dtf_cast_scale")
    return dst
