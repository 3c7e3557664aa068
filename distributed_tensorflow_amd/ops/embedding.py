"""Embedding lookup and sparse gradient apply (kernels K3 / K4).

Replaces TF1's ``embedding_lookup`` (``Gather`` on a PS-hosted partitioned
variable) and ``ScatterAdd`` / ``SparseApplyAdagrad`` on the PS (externals
X3/X4).  ``*_reference`` are the PyTorch definitions (CPU path + test oracle).
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import hip

SGD, ADAGRAD, ROWWISE_ADAGRAD = 0, 1, 2
_OPT = {"sgd": SGD, "adagrad": ADAGRAD, "rowwise_adagrad": ROWWISE_ADAGRAD}
_declared = False
_PLAN_KERNEL = os.environ.get("DTF_EMB_PLAN_KERNEL", "1") != "0"      # A/B switch: 0 = framework-op segment plan


def _lib():
    global _declared
    lib = hip.lib()
    if not _declared:
        P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
        hip.declare("dtf_embedding_gather", [P, I, P, P, I, L, I, I, L, F, P])
        hip.declare("dtf_embedding_sparse_apply", [P, I, P, P, I, P, P, P, P, P, P, P, L, I, I, L, I, F, F, F, F, P])
        hip.declare("dtf_embedding_sgd_atomic", [P, P, P, I, L, I, I, L, F, F, P])
        hip.declare("dtf_embedding_chunk_rows", [])
        hip.declare("dtf_embedding_plan_scratch_ints", [L], L)
        hip.declare("dtf_embedding_segment_plan", [P, L, I, P, P, P, P, P, P])
        hip.declare("dtf_embedding_route", [P, L, I, I, L, L, L, P, P, P, P, P, P, P])
        hip.declare("dtf_route_tile", [])
        hip.declare("dtf_sort_ids_temp_bytes", [L, L], L)
        hip.declare("dtf_sort_ids", [P, P, P, L, L, P, L, P])
        hip.declare("dtf_emb_split_ok", [L, I, I, I, I, I, L])
        hip.declare("dtf_emb_split_fwd", [P, P, P, P, L, I, I, I, I, L, P])
        hip.declare("dtf_emb_split_bwd", [P, P, P, L, I, I, L, P])
        _declared = True
    return lib


# ---------------------------------------------------------------------------
# K3
# ---------------------------------------------------------------------------

def gather_reference(table, idx, out_dtype=None, bag=1, mean=False):
    R, D = table.shape
    idx = idx.reshape(-1)
    valid = (idx >= 0) & (idx < R)
    rows = table[idx.clamp(0, R - 1)].float() * valid.unsqueeze(1)
    out = rows.view(-1, bag, D).sum(1)
    if mean:
        out = out / bag
    return out.to(out_dtype or table.dtype)


def embedding_gather(table, idx, out_dtype=None, bag=1, mean=False):
    """``out[b] = sum_j table[idx[b*bag + j]]`` -> ``[idx.numel() // bag, D]``.
    Ids outside ``[0, R)`` are padding and contribute zeros."""
    out_dtype = out_dtype or table.dtype
    if not table.is_cuda:
        return gather_reference(table, idx, out_dtype, bag, mean)
    assert table.dim() == 2 and table.is_contiguous() and idx.dtype == torch.int64
    idx = idx.reshape(-1).contiguous()
    assert idx.numel() % bag == 0
    R, D = table.shape
    nb = idx.numel() // bag
    out = torch.empty(nb, D, dtype=out_dtype, device=table.device)
    rc = _lib().dtf_embedding_gather(hip.ptr(table), hip.dt(table), hip.ptr(idx), hip.ptr(out), hip.dt(out), nb,
                                     int(bag), D, R, 1.0 / bag if mean else 1.0, hip.stream())
    hip.check(rc, "dtf_embedding_gather")
    return out


# ---------------------------------------------------------------------------
# K4
# ---------------------------------------------------------------------------

def sparse_apply_reference(table, acc, idx, grad, opt="sgd", lr=0.01, eps=1e-10, grad_scale=1.0, wd=0.0, bag=1):
    """Sum the gradients of duplicate ids, then one optimizer step per unique row."""
    R, D = table.shape
    idx = idx.reshape(-1)
    g_rows = grad.float().repeat_interleave(bag, 0) if bag > 1 else grad.float()
    valid = (idx >= 0) & (idx < R)
    idx, g_rows = idx[valid], g_rows[valid]
    uniq, inv = torch.unique(idx, return_inverse=True)
    g = torch.zeros(uniq.numel(), D, dtype=torch.float32, device=table.device).index_add_(0, inv, g_rows)
    w = table[uniq].float()
    g = g * grad_scale + wd * w
    if opt == "sgd":
        w = w - lr * g
    elif opt == "adagrad":
        h = acc[uniq] + g * g
        acc[uniq] = h
        w = w - lr * g * torch.rsqrt(h + eps)
    else:
        h = acc[uniq] + (g * g).mean(1)
        acc[uniq] = h
        w = w - lr * g * torch.rsqrt(h + eps).unsqueeze(1)
    table[uniq] = w.to(table.dtype)


_SORT_KERNEL = os.environ.get("DTF_EMB_SORT_KERNEL", "1") != "0"      # A/B switch: 0 = torch.sort (all 64 key bits)
_sort_ws = {}


def sort_ids(idx, rows=None):
    """Stable ascending sort of the row ids with their positions -> ``(sorted_ids, perm)``.  On the GPU with a known
    table height ``rows``: rocPRIM's radix sort over only the ``floor(log2 rows) + 1`` key bits a row id can have
    (``csrc/sort_kernels.hip``; 3 passes instead of 8 for config #3's 2.6 M rows); ids outside ``[0, rows)`` -- padding
    slots -- come out as the key ``rows``, after every valid id (the apply kernels skip both forms)."""
    flat = idx.reshape(-1)
    n = flat.numel()
    if not (_SORT_KERNEL and flat.is_cuda and rows and flat.dtype == torch.int64 and 0 < n < (1 << 31) - 4096):
        return torch.sort(flat, stable=True)
    flat = flat.contiguous()
    lib = _lib()
    need = int(lib.dtf_sort_ids_temp_bytes(n, int(rows)))
    if need < 0:
        return torch.sort(flat, stable=True)
    key = (flat.device, torch.cuda.current_stream(flat.device).cuda_stream)
    ws = _sort_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _sort_ws[key] = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=flat.device)
    out = torch.empty(2, n, dtype=torch.int64, device=flat.device)
    rc = lib.dtf_sort_ids(hip.ptr(flat), hip.ptr(out[0]), hip.ptr(out[1]), n, int(rows), hip.ptr(ws), ws.numel(),
                          hip.stream())
    hip.check(rc, "dtf_sort_ids")
    return out[0], out[1]


def segment_plan(idx, chunk, rows=None):
    """Sort ids and describe their runs with fixed-size tensors (no host sync):
    ``sorted_ids, perm, chunk_begin[n+1], seg_begin[n+1], seg_first_chunk[n+1], nchunks[1]``.
    On the GPU the run / chunk boundaries come from the plan kernels of ``embedding_kernels.hip``; ``rows`` (the
    table height) lets the sort skip the key bits no row id has."""
    n = idx.numel()
    dev = idx.device
    sid, perm = sort_ids(idx, rows)
    if idx.is_cuda and _PLAN_KERNEL and 0 < n < (1 << 31) - 4096:
        lib = _lib()
        scratch = torch.empty(int(lib.dtf_embedding_plan_scratch_ints(n)), dtype=torch.int32, device=dev)
        out = torch.empty(3 * (n + 1) + 1, dtype=torch.int64, device=dev)
        chunk_begin, seg_begin, seg_first_chunk = out[:n + 1], out[n + 1:2 * n + 2], out[2 * n + 2:3 * n + 3]
        nchunks = out[3 * n + 3:]
        rc = lib.dtf_embedding_segment_plan(hip.ptr(sid), n, int(chunk), hip.ptr(scratch), hip.ptr(chunk_begin),
                                            hip.ptr(seg_begin), hip.ptr(seg_first_chunk), hip.ptr(nchunks), hip.stream())
        hip.check(rc, "dtf_embedding_segment_plan")
        return sid, perm, chunk_begin, seg_begin, seg_first_chunk, nchunks
    return (sid, perm) + segment_plan_reference(sid, chunk)


def segment_plan_reference(sid, chunk):
    """The same boundaries with framework ops (CPU path, test oracle): ``chunk_begin, seg_begin,
    seg_first_chunk, nchunks`` of the sorted ids ``sid``."""
    n = sid.numel()
    dev = sid.device
    pos = torch.arange(n, device=dev)
    head = torch.ones(n, dtype=torch.bool, device=dev)
    head[1:] = sid[1:] != sid[:-1]
    # start position of every element's run: scatter the run heads, gather them back (a cummax over
    # 4e5 positions costs 1.2 ms in the framework's scan kernel -- 37 % of a Wide&Deep step)
    seg_id = torch.cumsum(head, 0) - 1
    big = torch.full((n + 1,), n, dtype=torch.int64, device=dev)
    seg_begin = big.clone().scatter_reduce_(0, seg_id, pos, "amin")
    seg_start_of = seg_begin[seg_id]
    chead = ((pos - seg_start_of) % chunk) == 0
    chunk_id = torch.cumsum(chead, 0) - 1
    chunk_begin = big.clone().scatter_reduce_(0, chunk_id, pos, "amin")
    seg_first_chunk = big.clone().scatter_reduce_(0, seg_id, chunk_id, "amin")
    nchunks = (chunk_id[-1:] + 1).contiguous()
    return chunk_begin, seg_begin, seg_first_chunk, nchunks


_ws = {}


def _workspace(device, numel):
    w = _ws.get(device)
    if w is None or w.numel() < numel:
        w = torch.empty(max(numel, 1 << 20), dtype=torch.float32, device=device)
        _ws[device] = w
    return w


def embedding_sparse_apply(table, acc, idx, grad, opt="sgd", lr=0.01, eps=1e-10, grad_scale=1.0, wd=0.0, bag=1,
                           atomic=False):
    """In-place sparse optimizer step on the rows ``idx`` of ``table``.

    ``grad`` is ``[idx.numel() // bag, D]`` (the gradient of the gather output).
    ``atomic=True`` selects the float-atomic form (SGD, fp32 table, ``wd == 0``)."""
    if not table.is_cuda:
        return sparse_apply_reference(table, acc, idx, grad, opt, lr, eps, grad_scale, wd, bag)
    lib = _lib()
    idx = idx.reshape(-1).contiguous()
    n = idx.numel()
    if n == 0:
        return
    R, D = table.shape
    assert grad.is_contiguous() and grad.shape == (n // bag, D) and idx.dtype == torch.int64
    if atomic:
        assert opt == "sgd" and table.dtype == torch.float32 and wd == 0.0
        rc = lib.dtf_embedding_sgd_atomic(hip.ptr(table), hip.ptr(idx), hip.ptr(grad), hip.dt(grad), n, D, int(bag), R,
                                          float(lr), float(grad_scale), hip.stream())
        hip.check(rc, "dtf_embedding_sgd_atomic")
        return
    chunk = int(lib.dtf_embedding_chunk_rows())
    sid, perm, chunk_begin, seg_begin, seg_first_chunk, nchunks = segment_plan(idx, chunk, rows=R)
    partial = _workspace(table.device, n * D)
    rc = lib.dtf_embedding_sparse_apply(
        hip.ptr(table), hip.dt(table), hip.ptr(acc), hip.ptr(grad), hip.dt(grad), hip.ptr(sid), hip.ptr(perm),
        hip.ptr(chunk_begin), hip.ptr(seg_begin), hip.ptr(seg_first_chunk), hip.ptr(nchunks), hip.ptr(partial),
        n, D, int(bag), R, _OPT[opt], float(lr), float(eps), float(grad_scale), float(wd), hip.stream())
    if rc == -31:
        raise ValueError("rowwise_adagrad needs D/8 to divide 64 (D in 8,16,32,...,512)")
    hip.check(rc, "dtf_embedding_sparse_apply")


# ---------------------------------------------------------------------------
# id routing for row-sharded tables (route_kernels.hip)
# ---------------------------------------------------------------------------

def route_reference(ids, W, cap, div=False, rows_per=0, total_rows=None):
    """Stable bucket-by-owner split with framework ops (CPU path, test oracle).  Returns
    ``send_ids [W*cap]`` (local row at the owner, -1 = empty), ``slot [n]`` (slot of request i, -1 = dropped or
    padding), ``inv [W*cap]`` (request held by a slot, -1 = empty), ``totals [W]``, ``dropped`` (0-dim)."""
    ids = ids.reshape(-1)
    n, dev = ids.numel(), ids.device
    total_rows = int(total_rows if total_rows is not None else (1 << 62))
    valid = (ids >= 0) & (ids < total_rows)
    if div:
        owner = torch.div(ids, rows_per, rounding_mode="floor")
        local = ids - owner * rows_per
    else:
        owner = ids % W
        local = torch.div(ids, W, rounding_mode="floor")
    owner = torch.where(valid, owner, torch.full_like(owner, W))
    onehot = owner.unsqueeze(1) == torch.arange(W, device=dev).unsqueeze(0)          # [n, W]
    rank = (torch.cumsum(onehot, 0) - 1)[torch.arange(n, device=dev), owner.clamp(max=W - 1)]
    totals = onehot.sum(0)
    keep = valid & (rank < cap)
    slot = torch.where(keep, owner * cap + rank, torch.full_like(owner, -1))
    send_ids = torch.full((W * cap,), -1, dtype=torch.int64, device=dev)
    inv = torch.full((W * cap,), -1, dtype=torch.int64, device=dev)
    send_ids[slot[keep]] = local[keep]
    inv[slot[keep]] = torch.arange(n, device=dev)[keep]
    dropped = (totals - cap).clamp(min=0).sum()
    return send_ids, slot, inv, totals, dropped


def embedding_route(ids, W, cap, div=False, rows_per=0, total_rows=None, overflow=None):
    """Device-side routing of global row ids to their owner ranks with a fixed per-owner capacity: no sort, no
    host synchronisation; ``overflow`` (int64 [1], device) accumulates the number of dropped requests."""
    ids = ids.reshape(-1)
    if not ids.is_cuda:
        send_ids, slot, inv, totals, dropped = route_reference(ids, W, cap, div, rows_per, total_rows)
        if overflow is not None:
            overflow += dropped
        return send_ids, slot, inv, totals
    lib = _lib()
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    n, dev = ids.numel(), ids.device
    tile = int(lib.dtf_route_tile())
    nblk = (n + tile - 1) // tile
    hist = torch.empty(max(1, nblk) * W, dtype=torch.int32, device=dev)
    out = torch.empty(W + 2 * W * cap + n, dtype=torch.int64, device=dev)
    totals, send_ids, inv, slot = out[:W], out[W:W + W * cap], out[W + W * cap:W + 2 * W * cap], out[W + 2 * W * cap:]
    if overflow is None:
        overflow = torch.zeros(1, dtype=torch.int64, device=dev)
    rc = lib.dtf_embedding_route(hip.ptr(ids), n, int(W), int(bool(div)), int(rows_per),
                                 int(total_rows if total_rows is not None else (1 << 62)), int(cap), hip.ptr(hist),
                                 hip.ptr(totals), hip.ptr(send_ids), hip.ptr(slot), hip.ptr(inv), hip.ptr(overflow),
                                 hip.stream())
    hip.check(rc, "dtf_embedding_route")
    return send_ids, slot, inv, totals


# ---------------------------------------------------------------------------
# dense-gradient embedding (an ordinary trainable nn.Embedding weight, e.g. BERT's tied word embeddings)
# ---------------------------------------------------------------------------
_DENSE_GRAD = os.environ.get("DTF_EMB_DENSE_GRAD", "1") != "0"      # A/B switch


class _DenseEmbedding(torch.autograd.Function):
    """``F.embedding`` whose weight gradient -- a dense ``[V, H]`` tensor -- is built by the K4 sort/segment kernels
    (``table += grad`` on a zero table: SGD with lr = -1; every row summed in fp32 in a fixed order and rounded once)
    or, for a handful of rows (token types), by a one-hot GEMM.  The framework's own backward
    (sort + ``sum_and_scatter`` + segment kernels) took 0.8 ms of a BERT-base step."""

    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.wshape, ctx.wdtype = weight.shape, weight.dtype
        return torch.nn.functional.embedding(ids, weight)

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        V, H = ctx.wshape
        g2 = dy.reshape(-1, H)
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        idx = ids.reshape(-1)
        if V <= 64:
            onehot = (idx.unsqueeze(1) == torch.arange(V, device=idx.device)).to(g2.dtype)
            return (onehot.t() @ g2).to(ctx.wdtype), None
        dw = torch.zeros(V, H, dtype=ctx.wdtype, device=dy.device)
        embedding_sparse_apply(dw, None, idx, g2, opt="sgd", lr=-1.0)
        return dw, None


def dense_embedding(ids, weight):
    """``F.embedding(ids, weight)``; see :class:`_DenseEmbedding` for the backward."""
    if (_DENSE_GRAD and weight.is_cuda and weight.dtype in (torch.bfloat16, torch.float32) and ids.dtype == torch.int64
            and weight.requires_grad and torch.is_grad_enabled() and weight.shape[1] % 8 == 0):
        return _DenseEmbedding.apply(weight, ids)
    return torch.nn.functional.embedding(ids, weight)


# ---------------------------------------------------------------------------
# K20: fused rows [B, F, D + 8] <-> deep-tower input [B, F*D + tail] + summed wide weights [B]
# ---------------------------------------------------------------------------
_SPLIT = os.environ.get("DTF_EMB_SPLIT_KERNEL", "1") != "0"      # A/B switch: 0 = tensor expressions


def split_rows_reference(e, dense, D, tail):
    """``e [B, F, D + pad]``, ``dense [B, nd]`` -> ``(h [B, F*D + tail], wide [B, 1] fp32)``: the deep vectors side by
    side, then the dense features and zeros up to ``tail`` columns; the wide weights (column ``D``) summed over F."""
    B, F_, W = e.shape
    nd = dense.shape[1]
    h = torch.empty(B, F_ * D + tail, dtype=e.dtype, device=e.device)
    h[:, :F_ * D].view(B, F_, D).copy_(e[:, :, :D])
    h[:, F_ * D:F_ * D + nd].copy_(dense)
    if tail > nd:
        h[:, F_ * D + nd:].zero_()
    return h, e[:, :, D].float().sum(1, keepdim=True)


def merge_rows_reference(d_h, d_wide, shape, D, dtype=None):
    """Gradient of :func:`split_rows_reference` w.r.t. ``e`` (either incoming gradient may be None)."""
    B, F_, W = shape
    src = d_h if d_h is not None else d_wide
    dt = dtype or src.dtype
    de = torch.empty(B, F_, W, dtype=dt, device=src.device)
    # three slice writes into one buffer (a concatenation along the short last axis runs at a third of the rate)
    if d_h is not None:
        de[:, :, :D].copy_(d_h[:, :F_ * D].view(B, F_, D))
    else:
        de[:, :, :D].zero_()
    if d_wide is not None:
        de[:, :, D:D + 1].copy_(d_wide.to(dt).reshape(B, 1, 1).expand(B, F_, 1))
    else:
        de[:, :, D:D + 1].zero_()
    if W - D - 1 > 0:
        de[:, :, D + 1:].zero_()
    return de


def split_kernel_eligible(e, dense, D, tail):
    if not (_SPLIT and e.is_cuda and e.dtype == torch.bfloat16 and dense.dtype == torch.bfloat16 and e.dim() == 3
            and e.is_contiguous() and dense.is_contiguous() and dense.shape[0] == e.shape[0]):
        return False
    B, F_, W = e.shape
    return int(_lib().dtf_emb_split_ok(B, F_, D, W, dense.shape[1], tail, F_ * D + tail)) == 0


def split_rows(e, dense, D, tail):
    """Kernel form of :func:`split_rows_reference` (one pass: ``rowsplit_kernels.hip``)."""
    if not split_kernel_eligible(e, dense, D, tail):
        return split_rows_reference(e, dense, D, tail)
    B, F_, W = e.shape
    ldh = F_ * D + tail
    h = torch.empty(B, ldh, dtype=e.dtype, device=e.device)
    wide = torch.empty(B, 1, dtype=torch.float32, device=e.device)
    rc = _lib().dtf_emb_split_fwd(hip.ptr(e), hip.ptr(dense), hip.ptr(h), hip.ptr(wide), B, F_, D, dense.shape[1], tail,
                                  ldh, hip.stream())
    hip.check(rc, "dtf_emb_split_fwd")
    return h, wide


def merge_rows(d_h, d_wide, shape, D, dtype=None):
    """Kernel form of :func:`merge_rows_reference` (``dtype``: that of the forward's ``e``)."""
    B, F_, W = shape
    src = d_h if d_h is not None else d_wide
    ok = (_SPLIT and src is not None and src.is_cuda and W == D + 8 and D % 8 == 0 and dtype in (None, torch.bfloat16)
          and (d_h is None or (d_h.dtype == torch.bfloat16 and d_h.dim() == 2 and d_h.stride(1) == 1
                               and d_h.stride(0) % 8 == 0 and d_h.stride(0) >= F_ * D and d_h.data_ptr() % 16 == 0))
          and (d_wide is None or (d_wide.dtype == torch.float32 and d_wide.numel() == B and d_wide.is_contiguous())))
    if not ok:
        return merge_rows_reference(d_h, d_wide, shape, D, dtype)
    de = torch.empty(B, F_, W, dtype=torch.bfloat16, device=src.device)
    rc = _lib().dtf_emb_split_bwd(hip.ptr(d_h), hip.ptr(d_wide), hip.ptr(de), B, F_, D,
                                  d_h.stride(0) if d_h is not None else 0, hip.stream())
    hip.check(rc, "dtf_emb_split_bwd")
    return de
