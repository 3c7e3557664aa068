"""Fused Linear + bias + activation (+ residual) (kernel K6; replaces TF's
``MatMul + BiasAdd + Relu/Gelu`` on the worker).

Forward is the hand-written gfx950 MFMA GEMM of ``csrc/gemm_kernels.hip`` with the whole
epilogue fused; backward computes ``dZ = dY * act'(Z)`` and the bias gradient in one
hand-written pass and leaves the two plain GEMMs (``dX = dZ W``, ``dW = dZ^T X``) to
hipBLASLt through ``torch.matmul`` (library GEMMs for plain products, by design).
CPU / fp32 inputs use ``F.linear`` -- also the oracle of the kernel tests.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip

ACT = {"none": 0, None: 0, "relu": 1, "gelu": 2, "gelu_tanh": 3}
_GELUS = ("gelu", "gelu_tanh")
_declared = False
_ws = {}


def _lib():
    global _declared
    lib = hip.lib()
    if not _declared:
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        hip.declare("dtf_gemm_bias_act", [P, P, P, I, P, P, P, I, I, I, I, P])
        hip.declare("dtf_act_fwd", [P, P, L, I, P])
        hip.declare("dtf_act_bwd_partial_floats", [L, I], L)
        hip.declare("dtf_act_bwd_bias", [P, P, P, P, I, P, L, I, I, P])
        hip.declare("dtf_gemm_tn_workspace", [I, I, I], L)
        hip.declare("dtf_gemm_tn", [P, P, P, I, P, I, P, I, I, I, P])
        _declared = True
    return lib


def _partial(dev, n):
    w = _ws.get(dev)
    if w is None or w.numel() < n:
        w = torch.empty(max(n, 1 << 18), dtype=torch.float32, device=dev)
        _ws[dev] = w
    return w


def linear_act_reference(x, weight, bias=None, act="none", residual=None):
    y = F.linear(x, weight, bias)
    if act == "relu":
        y = F.relu(y)
    elif act == "gelu":
        y = F.gelu(y)
    elif act == "gelu_tanh":
        y = F.gelu(y, approximate="tanh")
    if residual is not None:
        y = y + residual
    return y


import os as _os

_DISABLED = _os.environ.get("DTF_FUSED_LINEAR", "1") == "0"      # A/B switch for benchmarking only


def kernel_eligible(x, weight):
    return (not _DISABLED and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0
            and weight.is_contiguous())


def gemm_bias_act(x2, weight, bias, act, residual=None, save_preact=False):
    """Raw forward kernel call on 2-D operands -> (y, preact or None)."""
    M, K = x2.shape
    N = weight.shape[0]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
    pre = torch.empty_like(y) if save_preact else None
    rc = _lib().dtf_gemm_bias_act(hip.ptr(x2), hip.ptr(weight), hip.ptr(bias), hip.dt(bias) if bias is not None else 0,
                                  hip.ptr(residual), hip.ptr(y), hip.ptr(pre), M, N, K, ACT[act], hip.stream())
    hip.check(rc, "dtf_gemm_bias_act")
    return y, pre


_tn_ws = {}
_TN_DISABLED = _os.environ.get("DTF_GEMM_TN", "1") == "0"        # A/B switch
_TN_FUSE_BIAS = _os.environ.get("DTF_TN_FUSE_BIAS", "1") != "0"  # A/B switch
_PLAIN_LIB = _os.environ.get("DTF_LINEAR_PLAIN_LIB", "1") != "0"  # A/B switch: bias-only forward GEMMs through the library
# A/B switch: bias + GELU forward as library GEMM (bias epilogue, writes the pre-activation) + one activation pass
# instead of the fused K6 epilogue; wins where K6's tiles are epilogue-bound (small K, wide N: BERT's FFN1)
_ACT_LIB = _os.environ.get("DTF_LINEAR_ACT_LIB", "1") != "0"


def act_forward(z, act):
    """``act(z)`` for a contiguous bf16 tensor (kernel ``act_fwd_kernel``)."""
    y = torch.empty_like(z)
    rc = _lib().dtf_act_fwd(hip.ptr(z), hip.ptr(y), z.numel(), ACT[act], hip.stream())
    hip.check(rc, "dtf_act_fwd")
    return y


def gemm_tn_eligible(a, b):
    return (not _TN_DISABLED and a.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.dim() == 2
            and b.dim() == 2 and a.shape[0] == b.shape[0] and a.shape[0] % 32 == 0 and a.shape[0] >= 1024
            and a.shape[1] % 8 == 0 and b.shape[1] % 8 == 0 and a.is_contiguous() and b.is_contiguous())


def gemm_tn(a, b, out_dtype=torch.bfloat16, colsum_dtype=None):
    """``a.T @ b`` for row-major ``a [M, N]``, ``b [M, K]`` with the reduction over the rows (kernel K6T):
    the weight gradient ``dW = dZ^T X`` of a Linear layer / 1x1 convolution.  With ``colsum_dtype`` also
    returns ``a.sum(0)`` (the bias gradient) computed inside the same kernel: ``(dW, dbias)``."""
    M, N = a.shape
    K = b.shape[1]
    lib = _lib()
    n = int(lib.dtf_gemm_tn_workspace(M, N, K))
    ws = _tn_ws.get(a.device)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 22), dtype=torch.float32, device=a.device)
        _tn_ws[a.device] = ws
    out = torch.empty(N, K, dtype=out_dtype, device=a.device)
    cs = torch.empty(N, dtype=colsum_dtype, device=a.device) if colsum_dtype is not None else None
    rc = lib.dtf_gemm_tn(hip.ptr(a), hip.ptr(b), hip.ptr(out), hip.dt(out), hip.ptr(cs),
                         hip.dt(cs) if cs is not None else 0, hip.ptr(ws), M, N, K, hip.stream())
    hip.check(rc, "dtf_gemm_tn")
    return out if cs is None else (out, cs)


class _FusedLinear(torch.autograd.Function):
    """``fork=True`` also returns a pass-through alias of ``x`` for a second consumer (the residual branch
    of a transformer block): the two gradients of ``x`` are then summed by the dX GEMM itself
    (``dX = dZ . W + dSkip``, beta = 1) instead of a separate elementwise add.  ``own=True`` lets the GEMM
    accumulate in place into the incoming ``dSkip`` buffer -- only valid when nobody else reads that
    gradient tensor afterwards (see ``ops/conv.py``; true for the residual epilogue of a later
    ``FusedLinear`` / K10 LayerNorm chain, whose gradient is a fresh tensor consumed in stream order)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act, fork=False, own=False, defer_bias=False):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, weight.shape[0]).contiguous()
        if defer_bias:
            # X . W^T + residual by the library (beta = 1 on a copy-free out-of-place addmm); the bias vector is added
            # by the LayerNorm that consumes y (layer_norm_deferred_bias); its gradient still comes from backward()
            assert r2 is not None and act in ("none", None) and bias is not None
            y, pre = torch.addmm(r2, x2, weight.t()), None
        elif _PLAIN_LIB and act in ("none", None) and r2 is None and bias is not None and bias.dtype == x2.dtype:
            # nothing to fuse but the bias, which the GEMM library's own epilogue adds: 113 vs 136 us on BERT's
            # QKV projection (scripts/k6_epilogue_bench.py)
            y, pre = torch.addmm(bias, x2, weight.t()), None
        elif (_ACT_LIB and act in _GELUS and r2 is None and bias is not None and bias.dtype == x2.dtype
              and weight.shape[0] % 8 == 0 and weight.shape[0] >= 2 * weight.shape[1] and x2.shape[0] >= 4096):
            pre = torch.addmm(bias, x2, weight.t())
            y = act_forward(pre, act)
        else:
            y, pre = gemm_bias_act(x2, weight, bias, act, r2, save_preact=(act in _GELUS))
        z = y if act == "relu" else pre      # relu: mask = y > 0 (fused_linear keeps relu+residual unfused)
        ctx.save_for_backward(x2, weight, z)
        ctx.act, ctx.has_bias, ctx.has_res, ctx.shape = act, bias is not None, residual is not None, shp
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.fork, ctx.own = fork, own
        if fork:
            return y.view(*shp[:-1], weight.shape[0]), x.view_as(x)
        return y.view(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x2, weight, z = ctx.saved_tensors
        if dy is None:            # only the pass-through was used
            return dskip, None, None, None, None, None, None, None
        N = weight.shape[0]
        dy2 = dy.reshape(-1, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        M = dy2.shape[0]
        lib = _lib()
        dres = dy if ctx.has_res else None
        dbias = torch.empty(N, dtype=ctx.bias_dtype, device=dy.device) if ctx.has_bias else None
        # no activation: dZ = dY and the bias gradient rides in the weight-gradient kernel (K6T)
        fuse_dbias = (_TN_FUSE_BIAS and ctx.act in ("none", None) and ctx.has_bias and ctx.needs_input_grad[1] and N % 8 == 0
                      and weight.dtype == torch.bfloat16 and gemm_tn_eligible(dy2, x2))
        if fuse_dbias:
            dz = dy2
        elif ctx.act in ("relu",) + _GELUS or ctx.has_bias:
            if N % 8 == 0:
                need_dz = ctx.act in ("relu",) + _GELUS
                dz = torch.empty_like(dy2) if need_dz else None
                part = _partial(dy.device, int(lib.dtf_act_bwd_partial_floats(M, N)))
                rc = lib.dtf_act_bwd_bias(hip.ptr(dy2), hip.ptr(z), hip.ptr(dz), hip.ptr(dbias),
                                          hip.dt(dbias) if dbias is not None else 0, hip.ptr(part), M, N,
                                          ACT[ctx.act], hip.stream())
                hip.check(rc, "dtf_act_bwd_bias")
                if dz is None:
                    dz = dy2
            else:
                if ctx.act == "relu":
                    dz = dy2 * (z > 0)
                elif ctx.act == "gelu":
                    zf = z.float()
                    dz = (dy2.float() * (0.5 * (1 + torch.erf(zf * 0.7071067811865476))
                                         + zf * torch.exp(-0.5 * zf * zf) * 0.3989422804014327)).to(dy2.dtype)
                elif ctx.act == "gelu_tanh":
                    zf = z.float().requires_grad_()
                    with torch.enable_grad():
                        (g,) = torch.autograd.grad(F.gelu(zf, approximate="tanh").sum(), zf)
                    dz = (dy2.float() * g).to(dy2.dtype)
                else:
                    dz = dy2
                if dbias is not None:
                    dbias = dz.float().sum(0).to(ctx.bias_dtype)
        else:
            dz = dy2
        dx = None
        if ctx.needs_input_grad[0]:
            if dskip is None:
                dx = torch.matmul(dz, weight).view(ctx.shape)
            else:
                ds2 = dskip.reshape(-1, weight.shape[1])
                if ctx.own and ds2.is_contiguous() and ds2.data_ptr() == dskip.data_ptr():
                    dx = ds2.addmm_(dz, weight).view(ctx.shape)
                else:
                    dx = torch.addmm(ds2, dz, weight).view(ctx.shape)
        dw = None
        # the weight gradient is off the critical path: on the side stream (ops/streams.py) it shares the chip with the
        # next node's LayerNorm / attention / activation backward instead of running in series with them
        from . import streams as _streams

        # ... unless dZ IS the residual gradient this node returns (no activation + residual): the node that receives it
        # may accumulate onto it in place (own_skip_grad) while the side stream still reads it
        side = _streams.run_wgrad if not (ctx.has_res and dz.data_ptr() == dy.data_ptr()) else (lambda fn, *t: (_streams.join_into(torch.cuda.current_stream(dy.device), dy.device), fn())[1])
        if fuse_dbias:
            dw, dbias = side(lambda: gemm_tn(dz, x2, colsum_dtype=ctx.bias_dtype), weight, dz, x2)
        elif ctx.needs_input_grad[1]:
            if gemm_tn_eligible(dz, x2) and weight.dtype == torch.bfloat16:
                dw = side(lambda: gemm_tn(dz, x2), weight, dz, x2)          # K6T: transposed-operand MFMA GEMM, split over M
            else:
                dw = side(lambda: torch.matmul(dz.t(), x2), weight, dz, x2)
        return dx, dw, dbias, dres, None, None, None, None


def fused_linear(x, weight, bias=None, act="none", residual=None):
    """``act(x @ weight.T + bias) (+ residual)``; the HIP path for bf16 CUDA tensors."""
    if not kernel_eligible(x, weight) or (act == "relu" and residual is not None):
        return linear_act_reference(x, weight, bias, act, residual)
    return _FusedLinear.apply(x, weight, bias, residual, act)


def fused_linear_fork(x, weight, bias=None, act="none", own_skip_grad=False):
    """-> ``(fused_linear(x), x')`` with ``x'`` an alias of ``x`` for the residual branch."""
    if (not kernel_eligible(x, weight) or not torch.is_grad_enabled() or not x.requires_grad
            or _os.environ.get("DTF_LINEAR_FORK", "1") == "0"):
        return fused_linear(x, weight, bias, act), x
    return _FusedLinear.apply(x, weight, bias, None, act, True, own_skip_grad)


# A/B switch: projection + residual + LayerNorm with the GEMM in the library (beta = 1) and the bias inside K10
_RES_LN_LIB = _os.environ.get("DTF_LINEAR_RES_LN_LIB", "1") != "0"
# measured through the BERT step: +0.5 % with the FFN-output projection (K = 3072) on this path, -0.7 % when the
# attention-output projection (K = 768, where K6 and the library are on par) takes it too
_RES_LN_MIN_K = int(_os.environ.get("DTF_LINEAR_RES_LN_MIN_K", "2048"))


def linear_residual_layernorm(x, lin, residual, ln):
    """``ln(lin(x) + residual)`` (BERT's attention-output and FFN-output sublayers).

    Default path: the library GEMM accumulates onto the residual (``addmm``, beta = 1) and the projection's bias
    vector is added by the K10 LayerNorm kernels on the fly (``pre_bias``), so neither the fused K6 epilogue (whose
    K = 768 / 3072, N = 768 tiles are epilogue-bound: 187 vs 147 us on BERT's FFN2, ``scripts/k6_epilogue_bench.py``)
    nor a separate bias pass is needed.  The bias gradient is the column sum of the LayerNorm's input gradient and
    still rides in the K6T weight-gradient kernel of ``lin``.  Falls back to ``ln(lin(x, residual))``."""
    from . import layernorm as _ln

    ok = (_RES_LN_LIB and lin.act in ("none", None) and lin.bias is not None and kernel_eligible(x, lin.weight)
          and lin.in_features >= _RES_LN_MIN_K
          and len(ln.normalized_shape) == 1 and ln.elementwise_affine
          and _ln.kernel_eligible(residual, ln.weight, ln.bias) and lin.bias.dtype == ln.weight.dtype
          and lin.bias.dtype == x.dtype and lin.out_features % 8 == 0)
    if not ok:
        return ln(lin(x, residual=residual))
    y = _FusedLinear.apply(x, lin.weight, lin.bias, residual, lin.act, False, False, True)
    return _ln.layer_norm_deferred_bias(y, lin.bias, ln.weight, ln.bias, ln.eps)


class FusedLinear(nn.Linear):
    """``nn.Linear`` with the activation (and an optional residual) fused into the GEMM."""

    def __init__(self, in_features, out_features, bias=True, act="none"):
        super().__init__(in_features, out_features, bias)
        self.act = act

    def forward(self, x, residual=None):
        return fused_linear(x, self.weight, self.bias, self.act, residual)

    def forward_fork(self, x, own_skip_grad=False):
        return fused_linear_fork(x, self.weight, self.bias, self.act, own_skip_grad)

    def extra_repr(self):
        return super().extra_repr() + ", act=%s" % self.act
