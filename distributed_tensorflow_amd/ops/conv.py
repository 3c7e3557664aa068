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

from . import streams as _streams

# Route trace ("routing maze"): while ``ROUTE_LOG`` is a list every per-product kernel choice below
# appends ``(product, (cin, cout, kh, kw, stride), (N, H, W), choice)``.  ``route_trace()`` is the context manager,
# ``scripts/route_table.py`` prints the table of a model, ``tests/test_models_gpu.py`` pins ResNet-50's at the bench shape.
ROUTE_LOG = None


def _rt(product, weight, x, stride, choice):
    if ROUTE_LOG is not None:
        cout, cin, kh, kw = weight.shape
        ROUTE_LOG.append((product, (int(cin), int(cout), int(kh), int(kw), int(stride[0]) if stride else 1),
                          (int(x.shape[0]), int(x.shape[2]), int(x.shape[3])), choice))


class route_trace:
    """``with route_trace() as log: loss.backward()`` -> the kernel choice of every convolution product in the block."""

    def __enter__(self):
        global ROUTE_LOG
        self._old, ROUTE_LOG = ROUTE_LOG, []
        return ROUTE_LOG

    def __exit__(self, *exc):
        global ROUTE_LOG
        ROUTE_LOG = self._old
        return False

_MM_BWD_DATA_MIN_CIN = int(os.environ.get("DTF_CONV1X1_MM_DGRAD_MIN_CIN", "128"))
_MM_FWD_MIN_CIN = int(os.environ.get("DTF_CONV1X1_MM_FWD_MIN_CIN", "1024"))
_MM_FWD_SKIP = {int(c) for c in os.environ.get("DTF_CONV1X1_MM_FWD_SKIP", "").split(",") if c}
_MM_FWD_SKIP_MAX_COUT = int(os.environ.get("DTF_CONV1X1_MM_FWD_SKIP_MAX_COUT", "1000000"))
_MM_TUNED = os.environ.get("DTF_CONV1X1_MM_TUNED", "1") != "0"   # A/B switch: 0 = thresholds only


def _mm_fwd(x, weight):
    """Forward product of a 1x1 convolution through the GEMM library?  From ``_MM_FWD_MIN_CIN`` input channels
    always; below that only when the shipped per-shape solution choices are in use AND hold this exact GEMM
    (``utils/gemm_tuning.py``): with the library's default heuristics MIOpen wins there, with the recorded solutions
    the library does (ResNet-50: +3 %)."""
    cout, cin = weight.shape[0], weight.shape[1]
    if cin in _MM_FWD_SKIP and cout <= _MM_FWD_SKIP_MAX_COUT:
        return False
    if cin >= _MM_FWD_MIN_CIN:
        return True
    if _MM_TUNED:
        from ..utils import gemm_tuning

        shapes = gemm_tuning.tuned_gemm_shapes()
        return bool(shapes) and ("tn", cout, x.shape[0] * x.shape[2] * x.shape[3], cin) in shapes
    return False


# measured: NO -- the library has no good split-K solution for these (740-850 us per GEMM against ~150 us in MIOpen;
# ResNet-50 11 150 -> 9 070 images/s with all narrow 1x1 weight gradients on it); kept as a switch, off
_WRW_LIB = os.environ.get("DTF_CONV1X1_WRW_LIB", "0") != "0"
_WRW_LIB_ALL = os.environ.get("DTF_CONV1X1_WRW_LIB_ALL", "0") == "1"      # tuning runs: every narrow 1x1 weight gradient


def _mm_wrw(dy, weight):
    """Weight gradient ``dw = dy^T . x`` of a NARROW 1x1 convolution (the wide ones are K6T's) through the GEMM library
    instead of MIOpen?  Only when the shipped solution choices hold this exact GEMM (a split-K / stream-K solution:
    the reduction runs over all N*H*W pixels, the output is a few thousand elements)."""
    if _WRW_LIB_ALL:
        return True
    if not _MM_TUNED:
        return False
    from ..utils import gemm_tuning

    cout, cin = weight.shape[0], weight.shape[1]
    shapes = gemm_tuning.tuned_gemm_shapes()
    return bool(shapes) and ("nt", cin, cout, dy.shape[0] * dy.shape[2] * dy.shape[3]) in shapes


def _mm_dgrad(dy, weight):
    """Same question for the data gradient ``dx = dy . w``."""
    cout, cin = weight.shape[0], weight.shape[1]
    if cin >= _MM_BWD_DATA_MIN_CIN:
        return True
    if _MM_TUNED:
        from ..utils import gemm_tuning

        shapes = gemm_tuning.tuned_gemm_shapes()
        return bool(shapes) and ("nn", cin, dy.shape[0] * dy.shape[2] * dy.shape[3], cout) in shapes
    return False

# DTF_DETERMINISTIC (default 1): every weight gradient an own kernel can compute goes through it (K6T / K6T-conv / K14 /
# K19: fp32 slabs summed in a fixed order) instead of MIOpen's atomic-accumulating kernels -- ResNet-50's gradients are
# then bitwise reproducible run to run on the GPU (scripts/determinism_check.py at batch 256: 0 of 161 tensors differ;
# 3 with =0: the 64 -> 64 and 256 -> 128 1x1 layers at 56 x 56 and the first stride-2 3x3, which standalone are a little
# faster in MIOpen).  On the weight-gradient stream the difference does not reach the step: 20.33-20.40 vs 20.38-20.40 ms
# ResNet-50, 27.81 vs 27.75 Inception-v3, 47.83 vs 47.94 ResNet-152 (scripts/gpu_r3_ag.sh).
_DETERMINISTIC = os.environ.get("DTF_DETERMINISTIC", "1") != "0"
_DETERMINISTIC_STRICT = os.environ.get("DTF_DETERMINISTIC", "1") == "2"      # also the image-input layers (see _ConvFn)


def set_deterministic(level):
    """The job option ``deterministic=0|1|2`` (``train/jobspec.py``): what ``DTF_DETERMINISTIC`` sets at import, for a
    process that is already running (a warm rank has imported this module long before it learns its job)."""
    global _DETERMINISTIC, _DETERMINISTIC_STRICT
    level = int(level)
    if level not in (0, 1, 2):
        raise ValueError("deterministic must be 0, 1 or 2, not %r" % (level,))
    _DETERMINISTIC, _DETERMINISTIC_STRICT = level != 0, level == 2
    os.environ["DTF_DETERMINISTIC"] = str(level)
    return level
_TN_WRW_MIN_C = int(os.environ.get("DTF_CONV1X1_TN_MIN_C", "192"))      # K6T weight gradient when min(Cin, Cout) >= this
_TN_WRW = os.environ.get("DTF_CONV1X1_TN", "1") != "0"
_ENABLED = os.environ.get("DTF_CONV1X1_MM", "1") != "0"      # A/B switch: 0 = everything through MIOpen
_FORK = os.environ.get("DTF_CONV1X1_FORK", "1") != "0"        # A/B switch: 0 = autograd sums the shortcut grad


# K19 (csrc/conv1x1_wrw_kernels.hip): streaming weight gradient of the 64 <-> 256-channel 1x1 layers of the 56 x 56 stage
# (MIOpen: CK batched GEMM + workspace fill, 106-157 us per call at batch 256; K19: see profiles/r3_k19_wrw1x1.txt)
_WRW_STREAM = os.environ.get("DTF_CONV1X1_WRW_STREAM", "1") != "0"
_WRW_STREAM_MIN_ROWS = int(os.environ.get("DTF_CONV1X1_WRW_STREAM_MIN_ROWS", "131072"))
_wrw_stream_declared = False
_wrw_stream_ws = {}


def wrw_stream_ok(dy2, x2):
    """``dy2`` [M, Cout], ``x2`` [M, Cin] bf16 rows: can K19 take ``dW = dy2^T . x2``?"""
    if not (_WRW_STREAM and _ENABLED and dy2.is_cuda and dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16
            and dy2.dim() == 2 and x2.dim() == 2 and dy2.shape[0] == x2.shape[0]
            and (dy2.shape[1], x2.shape[1]) in ((256, 64), (64, 256))
            and dy2.shape[0] % 64 == 0 and dy2.shape[0] >= _WRW_STREAM_MIN_ROWS
            and dy2.is_contiguous() and x2.is_contiguous() and dy2.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0):
        return False
    from . import hip

    return hasattr(hip.lib(), "dtf_wrw1x1")


def wrw_stream(dy2, x2):
    """``dy2 [M, Cout]^T . x2 [M, Cin] -> [Cout, Cin]`` (bf16, fp32 accumulation, fixed summation order) on K19."""
    global _wrw_stream_declared
    import ctypes

    from . import hip

    lib = hip.lib()
    if not _wrw_stream_declared:
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        hip.declare("dtf_wrw1x1_ok", [L], I)
        hip.declare("dtf_wrw1x1_workspace_floats", [L], L)
        hip.declare("dtf_wrw1x1", [P, P, P, P, L, I, P])
        _wrw_stream_declared = True
    M, cout = dy2.shape
    cin = x2.shape[1]
    dev = dy2.device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _wrw_stream_ws.get(key)
    if ws is None:
        ws = _wrw_stream_ws[key] = torch.empty(int(lib.dtf_wrw1x1_workspace_floats(M)), dtype=torch.float32, device=dev)
    out = torch.empty(cout, cin, dtype=torch.bfloat16, device=dev)
    if cout == 256:         # A = dY (256 channels), B = X: C = dW
        rc = lib.dtf_wrw1x1(hip.ptr(dy2), hip.ptr(x2), hip.ptr(out), hip.ptr(ws), M, 0, hip.stream())
    else:                   # A = X (256 channels), B = dY: C = dW^T, stored transposed
        rc = lib.dtf_wrw1x1(hip.ptr(x2), hip.ptr(dy2), hip.ptr(out), hip.ptr(ws), M, 1, hip.stream())
    hip.check(rc, "dtf_wrw1x1")
    return out


def _wrw(dy, x, weight):
    """dW of a 1x1 stride-1 convolution: K6T (``dY^T . X`` with the pixels as reduction rows, no zero-fill /
    cast passes around it) for wide layers, K19 for the 64 <-> 256-channel layers of the 56 x 56 stage, MIOpen
    otherwise."""
    cout, cin = weight.shape[0], weight.shape[1]
    if weight.dtype == torch.bfloat16 and (cout, cin) in ((256, 64), (64, 256)) \
            and dy.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last):
        dy2, x2 = _rows(dy), _rows(x)
        if wrw_stream_ok(dy2, x2):
            _rt("wgrad", weight, x, None, "K19 streaming 1x1")
            return _as_filter1x1(wrw_stream(dy2, x2))
    # measured (scripts/wrw1x1_bench.py): K6T wins from 128 channels on both sides up to ~2.6e5 pixels (ResNet's 28x28
    # and Inception's 17x17 / 8x8 stages), from 192 channels beyond; the 56x56 layers are HBM-bound and MIOpen's
    if _TN_WRW and (_DETERMINISTIC or min(cin, cout) >= _TN_WRW_MIN_C
                    or (min(cin, cout) >= 128 and dy.shape[0] * dy.shape[2] * dy.shape[3] <= 262144)):
        from .linear import gemm_tn, gemm_tn_eligible

        dy2, x2 = _rows(dy), _rows(x)
        if gemm_tn_eligible(dy2, x2) and weight.dtype == torch.bfloat16:
            _rt("wgrad", weight, x, None, "K6T")
            return _as_filter1x1(gemm_tn(dy2, x2))
        if _DETERMINISTIC_STRICT and weight.dtype == torch.bfloat16 and dy2.is_cuda and dy2.dtype == torch.bfloat16:
            # K6T wants a row count that is a multiple of 32 and >= 1024 (e.g. 48 x 7 x 7 = 2352 is not): zero rows add
            # nothing to dY^T X, so strict mode pads both operands instead of handing the product to MIOpen
            M = dy2.shape[0]
            Mp = max(1024, (M + 31) // 32 * 32)
            dyp = torch.zeros(Mp, cout, dtype=dy2.dtype, device=dy2.device)
            xp = torch.zeros(Mp, cin, dtype=x2.dtype, device=x2.device)
            dyp[:M].copy_(dy2)
            xp[:M].copy_(x2)
            if gemm_tn_eligible(dyp, xp):
                _rt("wgrad", weight, x, None, "K6T over zero-padded rows")
                return _as_filter1x1(gemm_tn(dyp, xp))
    if _WRW_LIB and _mm_wrw(dy, weight):
        _rt("wgrad", weight, x, None, "hipBLASLt")
        return _as_filter1x1(torch.matmul(_rows(dy).t(), _rows(x)))
    _rt("wgrad", weight, x, None, "MIOpen")
    _, dw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [0, 0], [1, 1], False,
                                                   [0, 0], 1, [False, True, False])
    return dw


def _as_filter1x1(dw2):
    """[Cout, Cin] -> [Cout, Cin, 1, 1] with the strides of a channels-last parameter (Cin, 1, Cin, Cin): autograd's
    AccumulateGrad then keeps the tensor instead of copying it into the parameter's layout (28 copy kernels of ~5 us per
    ResNet-50 step with the plain ``view(cout, cin, 1, 1)``)."""
    cout, cin = dw2.shape
    return dw2.view(cout, 1, 1, cin).permute(0, 3, 1, 2)


def _rows(x):
    """NHWC-dense [N,C,H,W] -> contiguous [N*H*W, C] view (no copy)."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


# K18 (csrc/conv1x1_n64_kernels.hip): register-streaming product onto 64 channels (K = 64 / 256): the stage-1 squeeze
# layers' forward and the expand layers' data gradient (library: 125 us at K = 256, 64 / 76 us at K = 64)
_N64 = os.environ.get("DTF_CONV1X1_N64", "1") != "0"
_N64_MIN_ROWS = int(os.environ.get("DTF_CONV1X1_N64_MIN_ROWS", "65536"))
_n64_declared = False


def mm_n64_ok(x2, k, n):
    """``x2`` [M, k] bf16 rows, product onto ``n`` channels: can K18 take it?"""
    if not (_N64 and _ENABLED and n == 64 and k in (64, 256) and x2.is_cuda and x2.dtype == torch.bfloat16 and x2.dim() == 2
            and x2.shape[1] == k and x2.shape[0] % 16 == 0 and x2.shape[0] >= _N64_MIN_ROWS and x2.is_contiguous()
            and x2.data_ptr() % 16 == 0):
        return False
    from . import hip

    return hasattr(hip.lib(), "dtf_conv1x1_n64")


def mm_n64(x2, w2):
    """``x2 [M, K] . w2 [64, K]^T -> [M, 64]`` (bf16) on K18."""
    global _n64_declared
    import ctypes

    from . import hip

    lib = hip.lib()
    if not _n64_declared:
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        hip.declare("dtf_conv1x1_n64", [P, P, P, L, I, P])
        _n64_declared = True
    if not w2.is_contiguous():
        w2 = w2.contiguous()
    M, K = x2.shape
    assert tuple(w2.shape) == (64, K) and w2.dtype == torch.bfloat16
    y2 = torch.empty(M, 64, dtype=x2.dtype, device=x2.device)
    rc = lib.dtf_conv1x1_n64(hip.ptr(x2), hip.ptr(w2), hip.ptr(y2), M, K, hip.stream())
    hip.check(rc, "dtf_conv1x1_n64")
    return y2


def _fwd_n64(x, weight):
    """-> ``conv1x1(x, weight)`` on K18 or None."""
    cout, cin = weight.shape[0], weight.shape[1]
    if cout != 64 or weight.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
        return None
    x2 = _rows(x)
    if not mm_n64_ok(x2, cin, cout):
        return None
    N, _, H, W = x.shape
    return mm_n64(x2, weight.reshape(cout, cin)).view(N, H, W, cout).permute(0, 3, 1, 2)


def _dgrad_n64(dy, weight):
    """-> data gradient ``dy . w`` of a 1x1 convolution with 64 input channels on K18 or None (``dy`` NHWC-dense)."""
    cout, cin = weight.shape[0], weight.shape[1]
    if cin != 64 or weight.dtype != torch.bfloat16:
        return None
    dy2 = _rows(dy)
    if not mm_n64_ok(dy2, cout, cin):
        return None
    N, _, H, W = dy.shape
    return mm_n64(dy2, weight.reshape(cout, cin).t().contiguous()).view(N, H, W, cin).permute(0, 3, 1, 2)


class _Conv1x1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, mm_fwd=False):
        ctx.save_for_backward(x, weight)
        cout, cin = weight.shape[0], weight.shape[1]
        y = _fwd_n64(x, weight)
        if y is not None:
            _rt("fwd", weight, x, None, "K18 streaming onto 64 channels")
            return y
        if mm_fwd or _mm_fwd(x, weight):
            _rt("fwd", weight, x, None, "hipBLASLt")
            y2 = torch.matmul(_rows(x), weight.reshape(cout, cin).t())
            N, _, H, W = x.shape
            return y2.view(N, H, W, cout).permute(0, 3, 1, 2)
        if _DETERMINISTIC_STRICT and conv_fwd_eligible(x, weight, (1, 1), (0, 0)):
            # a shape neither library has a recorded choice for (any batch size but the tuned ones) gets MIOpen's
            # heuristic pick, which may accumulate with atomics (ResNet-50 at batch 32: the forward itself differed
            # run to run); strict mode keeps such products on the own pipeline
            return conv_fwd(x, weight, (1, 1), (0, 0))
        _rt("fwd", weight, x, None, "MIOpen")
        return F.conv2d(x, weight)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        need_dx, need_dw = ctx.needs_input_grad[:2]
        if need_dx:
            dx = _dgrad_n64(dy, weight)
            need_dx = dx is None
            if not need_dx:
                _rt("dgrad", weight, x, None, "K18 streaming onto 64 channels")
        if need_dx and _mm_dgrad(dy, weight):
            _rt("dgrad", weight, x, None, "hipBLASLt")
            N, _, H, W = x.shape
            dx = torch.matmul(_rows(dy), weight.reshape(cout, cin)).view(N, H, W, cin).permute(0, 3, 1, 2)
            need_dx = False
        if need_dx and _DETERMINISTIC_STRICT:
            wt = weight.reshape(cout, cin).t().contiguous().view(cin, 1, 1, cout).permute(0, 3, 1, 2)    # [Cin, Cout, 1, 1]
            if conv_fwd_eligible(dy, wt, (1, 1), (0, 0)):
                dx = conv_fwd(dy, wt, (1, 1), (0, 0), _product="dgrad")
                need_dx = False
        if need_dx:
            _rt("dgrad", weight, x, None, "MIOpen")
            dx, _, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [0, 0], [1, 1], False,
                                                           [0, 0], 1, [True, False, False])
        if need_dw:
            dw = _streams.run_wgrad(lambda: _wrw(dy, x, weight), weight, dy, x)
        return dx, dw, None


# BatchNorm statistics in the K6 GEMM epilogue.  Off by default: measured on ResNet-50 batch 256
# (scripts/conv1x1_stats_bench.py) the saved statistics pass (30-56 us per expand layer) is eaten by K6 being
# 15-50 us slower than MIOpen's kernels on these memory-bound shapes (9 650 vs 9 680 images/s end to end).
_STATS = os.environ.get("DTF_CONV1X1_STATS", "0") != "0"
_STATS_MIN_COUT = int(os.environ.get("DTF_CONV1X1_STATS_MIN_COUT", "256"))
_STATS_K13 = os.environ.get("DTF_CONV1X1_STATS_K13", "1") != "0"
_stats_declared = False


class _Conv1x1StatsFn(torch.autograd.Function):
    """1x1 convolution on K6 whose epilogue also reduces the output over the pixels: ``(y, partials)`` with
    ``partials`` = fp32 ``[2 * ceil(M / 256), 2, Cout]`` column sums / sums of squares, which the following
    training-mode BatchNorm (``FusedBatchNormAct(..., stats=partials)``) finalizes instead of reading ``y`` once
    more -- the statistics pass is 2 B/element of pure HBM traffic on the widest activations of a ResNet."""

    @staticmethod
    def forward(ctx, x, weight):
        global _stats_declared
        import ctypes

        from . import hip

        lib = hip.lib()
        if not _stats_declared:
            P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
            hip.declare("dtf_gemm_stats_blocks", [L], L)
            hip.declare("dtf_gemm_stats", [P, P, P, P, I, I, I, P])
            _stats_declared = True
        ctx.save_for_backward(x, weight)
        _rt("fwd", weight, x, None, "K6 / K13 pipeline + BN statistics")
        if _STATS_K13:          # the same pipeline through the convolution entry point: tile picked per problem
            y, part = conv_fwd(x, weight, (1, 1), (0, 0), stats=True, _traced=True)
            ctx.mark_non_differentiable(part)
            return y, part
        cout, cin = weight.shape[0], weight.shape[1]
        N, _, H, W = x.shape
        x2 = _rows(x)
        M = x2.shape[0]
        y2 = torch.empty(M, cout, dtype=x.dtype, device=x.device)
        part = torch.empty(int(lib.dtf_gemm_stats_blocks(M)), 2, cout, dtype=torch.float32, device=x.device)
        rc = lib.dtf_gemm_stats(hip.ptr(x2), hip.ptr(weight), hip.ptr(y2), hip.ptr(part), M, cout, cin, hip.stream())
        hip.check(rc, "dtf_gemm_stats")
        ctx.mark_non_differentiable(part)
        return y2.view(N, H, W, cout).permute(0, 3, 1, 2), part

    @staticmethod
    def backward(ctx, dy, _dpart):
        return _Conv1x1Fn.backward(ctx, dy)[:2]


# K17 (csrc/conv1x1_stats_kernels.hip): streaming 1x1 kernel with the statistics epilogue for the few-input-channel expand
# layers (64 -> 256): replaces the library GEMM (107 us) + the K7 statistics pass over its 411 MB output (75 us)
_K17 = os.environ.get("DTF_CONV1X1_K17", "1") != "0"
_k17_declared = False


class _Conv1x1K17Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        global _k17_declared
        import ctypes

        from . import hip

        lib = hip.lib()
        if not _k17_declared:
            P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
            hip.declare("dtf_conv1x1_stats_blocks", [L, I, I], L)
            hip.declare("dtf_conv1x1_stats", [P, P, P, P, P, L, I, I, P])
            _k17_declared = True
        ctx.save_for_backward(x, weight)
        ctx.set_materialize_grads(False)
        _rt("fwd", weight, x, None, "K17 streaming 64 -> 256 + BN statistics")
        cout, cin = weight.shape[0], weight.shape[1]
        N, _, H, W = x.shape
        M = N * H * W
        dev = x.device
        z = _zero_page.get(dev)
        if z is None:
            z = torch.zeros(64, dtype=torch.bfloat16, device=dev)
            _zero_page[dev] = z
        y = torch.empty((N, H, W, cout), dtype=x.dtype, device=dev)
        part = torch.empty(int(lib.dtf_conv1x1_stats_blocks(M, cin, cout)), 2, cout, dtype=torch.float32, device=dev)
        rc = lib.dtf_conv1x1_stats(hip.ptr(x), hip.ptr(weight), hip.ptr(z), hip.ptr(y), hip.ptr(part), M, cin, cout, hip.stream())
        hip.check(rc, "dtf_conv1x1_stats")
        ctx.mark_non_differentiable(part)
        return y.permute(0, 3, 1, 2), part

    @staticmethod
    def backward(ctx, dy, _dpart=None):
        return _Conv1x1Fn.backward(ctx, dy)[:2]


def _k17_ok(x, weight):
    if not (_K17 and _ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4
            and tuple(weight.shape[1:]) == (64, 1, 1) and weight.shape[0] == 256 and x.shape[1] == 64
            and x.is_contiguous(memory_format=torch.channels_last) and weight.is_contiguous(memory_format=torch.channels_last)
            and x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0 and torch.is_grad_enabled()):
        return False
    from . import hip

    lib = hip.lib()
    if not hasattr(lib, "dtf_conv1x1_stats"):
        return False
    return x.shape[0] * x.shape[2] * x.shape[3] >= 65536          # a streaming kernel: large pixel counts only


k17_ok = _k17_ok


def conv1x1_stats(x, weight):
    """-> ``(conv1x1(x, weight), partials or None)``; see :class:`_Conv1x1StatsFn`."""
    if _k17_ok(x, weight):
        return _Conv1x1K17Fn.apply(x, weight)
    if (_STATS and _eligible(x, weight) and weight.dtype == torch.bfloat16 and weight.shape[0] >= _STATS_MIN_COUT
            and weight.shape[0] % 8 == 0 and weight.shape[1] % 32 == 0 and weight.is_contiguous(memory_format=torch.channels_last)
            and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 31) - 256):
        return _Conv1x1StatsFn.apply(x, weight)
    return conv1x1(x, weight), None


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
        y = _fwd_n64(x, weight)
        if y is not None:
            _rt("fwd", weight, x, None, "K18 streaming onto 64 channels")
        elif _mm_fwd(x, weight):
            _rt("fwd", weight, x, None, "hipBLASLt")
            N, _, H, W = x.shape
            y = torch.matmul(_rows(x), weight.reshape(cout, cin).t()).view(N, H, W, cout).permute(0, 3, 1, 2)
        elif _DETERMINISTIC_STRICT and conv_fwd_eligible(x, weight, (1, 1), (0, 0)):
            y = conv_fwd(x, weight, (1, 1), (0, 0))
        else:
            _rt("fwd", weight, x, None, "MIOpen")
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
            _rt("dgrad", weight, x, None, "hipBLASLt" if dskip is None else "hipBLASLt, shortcut gradient added by beta = 1")
            if dskip is None:
                dx2 = torch.matmul(_rows(dy), w2)
            else:
                if not dskip.is_contiguous(memory_format=torch.channels_last):
                    dskip = dskip.contiguous(memory_format=torch.channels_last)
                dx2 = _rows(dskip).addmm_(_rows(dy), w2) if ctx.own else torch.addmm(_rows(dskip), _rows(dy), w2)
            dx = dx2.view(N, H, W, cin).permute(0, 3, 1, 2)
        if need_dw:
            dw = _streams.run_wgrad(lambda: _wrw(dy, x, weight), weight, dy, x)
        return dx, dw, None


class _SubsampleFork(torch.autograd.Function):
    """``(x[:, :, ::s, ::s] as a dense tensor, x)`` for a strided 1x1 convolution next to another consumer of ``x``
    (the shortcut of a ResNet stage's first block).  A strided 1x1 convolution only ever sees every s-th pixel, so it
    is the dense 1x1 GEMM of the subsampled input; its data gradient lives on those pixels only.  Backward adds it
    there IN PLACE into the gradient that comes back through the pass-through (a fresh tensor from the other
    consumer's data-gradient GEMM) -- instead of a library backward-data kernel writing a full-resolution, 3/4-zero
    tensor that the other consumer's GEMM then has to read and accumulate into."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        if _subsample_kernel_ok(x):
            N, C, H, W = x.shape
            out = torch.empty((N, (H + s - 1) // s, (W + s - 1) // s, C), dtype=x.dtype, device=x.device)
            _subsample(x, out, s, 0)
            return out.permute(0, 3, 1, 2), x.view_as(x)
        return x[:, :, ::s, ::s].contiguous(memory_format=torch.channels_last), x.view_as(x)

    @staticmethod
    def backward(ctx, dxs, dskip):
        s = ctx.s
        if dskip is None:
            raise RuntimeError("subsample_fork: the pass-through output must take part in the loss")
        if dxs is not None:
            if _subsample_kernel_ok(dskip) and _subsample_kernel_ok(dxs):
                _subsample(dskip, dxs.permute(0, 2, 3, 1), s, 1)
            else:
                dskip[:, :, ::s, ::s].add_(dxs)
        return dskip, None


_SUBSAMPLE_KERNEL = os.environ.get("DTF_SUBSAMPLE_KERNEL", "1") != "0"      # A/B switch: 0 = framework strided copy / add
_ss_declared = False


def _subsample_kernel_ok(x):
    return (_SUBSAMPLE_KERNEL and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0)


def _subsample(big, small, s, add):
    """``small = big[:, :, ::s, ::s]`` (add = 0) or ``big[:, :, ::s, ::s] += small`` (add = 1); ``big`` is an NCHW-shaped
    channels-last tensor, ``small`` a dense [N, OH, OW, C] one."""
    global _ss_declared
    import ctypes

    from . import hip

    lib = hip.lib()
    if not _ss_declared:
        P, I = ctypes.c_void_p, ctypes.c_int
        hip.declare("dtf_subsample_nhwc", [P, P, I, I, I, I, I, I, P])
        _ss_declared = True
    N, C, H, W = big.shape
    hip.check(lib.dtf_subsample_nhwc(hip.ptr(big), hip.ptr(small), N, H, W, C, int(s), int(add), hip.stream()),
              "dtf_subsample_nhwc")


def subsample_fork(x, stride):
    """-> ``(x[:, :, ::stride, ::stride]`` dense, ``x')``; route the other consumer of ``x`` through ``x'``."""
    return _SubsampleFork.apply(x, int(stride))


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


def conv1x1(x, weight, mm_fwd=False):
    """``mm_fwd``: forward through the GEMM library whatever ``Cin`` (shapes the convolution library has no
    tuning for, e.g. the concatenated branch heads of an Inception block)."""
    if (_ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)
            and weight.shape[2] == 1 and weight.shape[3] == 1):
        return _Conv1x1Fn.apply(x, weight, mm_fwd)
    _rt("fwd + bwd", weight, x, None, "framework conv2d (MIOpen)")
    return F.conv2d(x, weight)


class Conv1x1(nn.Conv2d):
    """``nn.Conv2d(cin, cout, 1, bias=False)`` with the per-product kernel choice above."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, 1, bias=False)

    def forward(self, x):
        return conv1x1(x, self.weight)

    def forward_fork(self, x, own_skip_grad=False):
        return conv1x1_fork(x, self.weight, own_skip_grad)

    def forward_stats(self, x):
        """-> ``(self(x), stats)`` for a following ``FusedBatchNormAct(y, ..., stats=stats)``."""
        return conv1x1_stats(x, self.weight)


# ---------------------------------------------------------------------------
# general KH x KW convolutions: weight gradient through K6T with an implicit im2col gather
# ---------------------------------------------------------------------------
_wrw_declared = False
_wrw_ws = {}
_zero_page = {}
_wrw_tab = {}
_CONV_WRW = os.environ.get("DTF_CONV_WRW", "1") != "0"          # A/B switch
# Measured against MIOpen's tuned solvers (scripts/conv_wrw_bench.py): K6T wins from 160 channels on both sides
# (ResNet stage 3/4 3x3, Inception's 8x8 blocks) and loses on the narrow, spatially large layers.
_CONV_WRW_MIN_CH = int(os.environ.get("DTF_CONV_WRW_MIN_CH", "160"))


def _wrw_lib():
    global _wrw_declared
    import ctypes

    from . import hip

    lib = hip.lib()
    if not _wrw_declared:
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        hip.declare("dtf_conv_wrw_workspace", [L, I, I], L)
        hip.declare("dtf_conv_wrw_table_ints", [L], L)
        hip.declare("dtf_conv_wrw_table", [P, I, I, I, I, I, I, I, I, P])
        hip.declare("dtf_conv_wrw", [P, P, P, I, P, P, P, I, I, I, I, I, I, I, I, I, P])
        hip.declare("dtf_conv_wrw_direct_ok", [I, I, I, I, I, I, I, I, I, I])
        hip.declare("dtf_conv_wrw_direct_workspace", [I, I, I, I, I], L)
        hip.declare("dtf_conv_wrw_direct", [P, P, P, I, P, P, I, I, I, I, I, P])
        hip.declare("dtf_conv_wrw_direct_workspace2", [I, I, I, I, I, I], L)
        hip.declare("dtf_conv_wrw_direct2", [P, P, P, I, P, P, I, I, I, I, I, I, P])
        _wrw_declared = True
    return lib


def conv_wrw_eligible(dy, x, weight, stride, padding, dilation=(1, 1), groups=1, min_k=256):
    # min_k: below one 256-wide K tile the kernel works (columns past K read the zero page) but wastes most of its tile
    return (_CONV_WRW and x.is_cuda and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and x.dim() == 4 and groups == 1 and tuple(dilation) == (1, 1)
            and stride[0] == stride[1] and x.shape[1] % 8 == 0 and weight.shape[0] % 8 == 0
            and weight.shape[1] * weight.shape[2] * weight.shape[3] >= min_k
            and x.is_contiguous(memory_format=torch.channels_last) and dy.is_contiguous(memory_format=torch.channels_last)
            and weight.is_contiguous(memory_format=torch.channels_last)
            and dy.shape[0] * dy.shape[2] * dy.shape[3] < (1 << 24))


# K14: the direct kernel for the narrow 3x3 stride-1 layers (csrc/conv_wrw_direct.hip); channels in multiples of 64 up to
# DTF_CONV_WRW_DIRECT_MAX_CH (wider layers have enough output tiles for K6T)
_CONV_WRW_DIRECT = os.environ.get("DTF_CONV_WRW_DIRECT", "1") != "0"
_CONV_WRW_DIRECT_MAX_CH = int(os.environ.get("DTF_CONV_WRW_DIRECT_MAX_CH", "128"))


def conv_wrw_direct_ok(x, weight, stride, padding):
    if not (_CONV_WRW_DIRECT and _CONV_WRW and x.is_cuda):
        return False
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or tuple(stride) != (1, 1) or tuple(padding) not in ((1, 1), (0, 0)) or cin % 32 or cout % 32:
        return False
    if max(cin, cout) > _CONV_WRW_DIRECT_MAX_CH or x.shape[1] != cin:
        return False
    p = int(padding[0])
    return bool(_wrw_lib().dtf_conv_wrw_direct_ok(x.shape[0], x.shape[2], x.shape[3], cin, cout, kh, kw, 1, p, p))


def conv_wrw_profitable(weight, x=None, stride=(1, 1), padding=(1, 1)):
    """Where the module form switches to K6T / K14 on its own (``conv_wrw`` itself runs any eligible shape)."""
    if _DETERMINISTIC or min(weight.shape[0], weight.shape[1]) >= _CONV_WRW_MIN_CH:
        return True
    return x is not None and conv_wrw_direct_ok(x, weight, stride, padding)


def conv_wrw(dy, x, weight, stride, padding):
    """``dW`` of ``conv2d(x, weight, stride, padding)`` for NHWC bf16 tensors (K6T with per-lane im2col source
    addresses: the patch matrix is never materialised, no fp32 workspace zero-fill / cast passes, no atomics:
    bitwise reproducible).  Returns a channels-last ``[Cout, Cin, KH, KW]`` tensor like the weight."""
    from . import hip

    lib = _wrw_lib()
    NB, Cin, H, W = x.shape
    _, Cout, OH, OW = dy.shape
    KH, KW = weight.shape[2], weight.shape[3]
    dev = x.device
    if conv_wrw_direct_ok(x, weight, stride, padding):
        pad = int(padding[0])
        n = int(lib.dtf_conv_wrw_direct_workspace2(NB, H, W, Cin, Cout, pad))
        ws = _wrw_ws.get(dev)
        if ws is None or ws.numel() < n:
            ws = torch.empty(max(n, 1 << 22), dtype=torch.float32, device=dev)
            _wrw_ws[dev] = ws
        z = _zero_page.get(dev)
        if z is None:
            z = torch.zeros(64, dtype=torch.bfloat16, device=dev)
            _zero_page[dev] = z
        dw = torch.empty((Cout, KH, KW, Cin), dtype=torch.bfloat16, device=dev)
        rc = lib.dtf_conv_wrw_direct2(hip.ptr(dy), hip.ptr(x), hip.ptr(dw), hip.dt(dw), hip.ptr(ws), hip.ptr(z), NB, H, W,
                                      Cin, Cout, pad, hip.stream())
        hip.check(rc, "dtf_conv_wrw_direct")
        return dw.permute(0, 3, 1, 2)
    n = int(lib.dtf_conv_wrw_workspace(NB * OH * OW, Cout, KH * KW * Cin))
    ws = _wrw_ws.get(dev)
    if ws is None or ws.numel() < n:
        ws = torch.empty(max(n, 1 << 22), dtype=torch.float32, device=dev)
        _wrw_ws[dev] = ws
    z = _zero_page.get(dev)
    if z is None:
        z = torch.zeros(64, dtype=torch.bfloat16, device=dev)
        _zero_page[dev] = z
    geom = (dev, NB, H, W, OH, OW, int(stride[0]), int(padding[0]), int(padding[1]))
    tab = _wrw_tab.get(geom)
    if tab is None:           # output pixel -> patch origin, once per layer geometry
        tab = torch.empty(int(lib.dtf_conv_wrw_table_ints(NB * OH * OW)), dtype=torch.int32, device=dev)
        hip.check(lib.dtf_conv_wrw_table(hip.ptr(tab), *geom[1:], hip.stream()), "dtf_conv_wrw_table")
        _wrw_tab[geom] = tab
    dw = torch.empty((Cout, KH, KW, Cin), dtype=torch.bfloat16, device=dev)
    rc = lib.dtf_conv_wrw(hip.ptr(dy), hip.ptr(x), hip.ptr(dw), hip.dt(dw), hip.ptr(ws), hip.ptr(z), hip.ptr(tab), NB, H, W,
                          Cin, OH, OW, Cout, KH, KW, hip.stream())
    hip.check(rc, "dtf_conv_wrw")
    return dw.permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------
# K21: weight gradient of image-input layers (Cin * KH * KW <= 32: 3x3 on RGB, 5x5 on grey-scale) as patch matrix + K6T
# (csrc/im2col_kernels.hip).  MIOpen's kernel for these layers accumulates with atomics; this one is bitwise reproducible
# and part of the default deterministic mode (Inception-v3's 3 -> 32 stem layer: the one gradient of the model that
# differed run to run).
# ---------------------------------------------------------------------------
_IMG_WRW = os.environ.get("DTF_CONV_IMAGE_WRW", "1") != "0"
_IMG_WRW_GROUP = int(os.environ.get("DTF_CONV_IMAGE_WRW_GROUP", "8"))       # A/B switch: 1 = plain [M, 32] operands
_img_declared = False


def _img_lib():
    global _img_declared
    import ctypes

    from . import hip

    lib = hip.lib()
    if not _img_declared and hasattr(lib, "dtf_im2col_small"):
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        hip.declare("dtf_im2col_small_ok", [L, I, I, I, I, I, I, I, I], I)
        hip.declare("dtf_im2col_small", [P, P, L, I, I, I, I, I, I, I, I, L, P])
        _img_declared = True
    return lib


def image_wrw_eligible(dy, x, weight, stride, padding):
    cout, cin, kh, kw = weight.shape
    if not (_IMG_WRW and _CONV_WRW and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and cin == x.shape[1] and cin * kh * kw <= 32 and cout % 8 == 0
            and stride[0] == stride[1] and x.is_contiguous(memory_format=torch.channels_last)
            and dy.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0
            and dy.shape[0] * dy.shape[2] * dy.shape[3] < (1 << 31) - (1 << 26)):
        return False
    lib = _img_lib()
    return hasattr(lib, "dtf_im2col_small") and lib.dtf_im2col_small_ok(x.shape[0], x.shape[2], x.shape[3], cin, kh, kw,
                                                                         int(stride[0]), int(padding[0]), int(padding[1])) == 0


def image_wrw(dy, x, weight, stride, padding):
    """``dW`` of ``conv2d(x, weight, stride, padding)`` for an input with a few channels (``Cin * KH * KW <= 32``): the patch
    matrix ``[pixels, 32]`` written once by K21, then ``dY^T . P`` on K6T.  Channels-last ``[Cout, Cin, KH, KW]`` result."""
    from . import hip
    from .linear import gemm_tn

    lib = _img_lib()
    cout, cin, kh, kw = weight.shape
    N, _, H, W = x.shape
    OH, OW = dy.shape[2], dy.shape[3]
    M = N * OH * OW
    Mp = max(1024, (M + 31) // 32 * 32)         # K6T: whole 32-row reduction tiles
    P = torch.empty(Mp, 32, dtype=torch.bfloat16, device=x.device)
    rc = lib.dtf_im2col_small(hip.ptr(x), hip.ptr(P), N, H, W, cin, kh, kw, int(stride[0]), int(padding[0]), int(padding[1]), Mp,
                              hip.stream())
    hip.check(rc, "dtf_im2col_small")
    dy2 = _rows(dy)
    if Mp != M:
        dyp = torch.zeros(Mp, cout, dtype=dy.dtype, device=dy.device)
        dyp[:M].copy_(dy2)
        dy2 = dyp
    if _IMG_WRW_GROUP > 1 and Mp == M and M % (32 * _IMG_WRW_GROUP) == 0 and cout * _IMG_WRW_GROUP <= 512:
        # K6T streams two 32-column operands at half its rate (a 256-wide K tile of which 32 columns are real: 277 us for
        # the 728 MB of Inception-v3's stem layer).  Rows taken G at a time -- dY as [M / G, G * Cout], P as [M / G, G * 32],
        # both free views -- give it full-width tiles; the wanted product is the sum of the G diagonal blocks of the
        # [G * Cout, G * 32] result (the off-diagonal blocks pair different pixels: G times the arithmetic, which the
        # matrix cores do not notice next to the HBM stream).  fp32 blocks, summed in a fixed order.
        G = _IMG_WRW_GROUP
        big = gemm_tn(dy2.view(M // G, G * cout), P.view(M // G, G * 32), out_dtype=torch.float32)
        g = torch.diagonal(big.view(G, cout, G, 32), dim1=0, dim2=2).sum(-1).to(torch.bfloat16)      # [Cout, 32]
    else:
        g = gemm_tn(dy2, P)                      # [Cout, 32]: columns (r * KW + s) * Cin + c, the rest zero
    # contiguous(): the 27 of 32 columns as a dense [Cout, KH, KW, Cin] tensor -- a strided view here would make autograd
    # clone the gradient into the parameter's layout on ITS stream (see streams._in_param_layout)
    return g[:, :cin * kh * kw].contiguous().view(cout, kh, kw, cin).permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------
# K13: forward convolution as an implicit GEMM on the K6 LDS-DMA pipeline (csrc/conv_kernels.hip)
# ---------------------------------------------------------------------------
_fwd_declared = False
# "auto": the layers where K13 measured faster than MIOpen's tuned kernel (_k13_profitable); "all": every eligible
# layer; "0": off.  A/B table: profiles/r2_conv_fwd_k13_vs_miopen.txt
_CONV_FWD = os.environ.get("DTF_CONV_FWD", "auto")
# measured at batch 256 (scripts/conv_fwd_bench.py, profiles/r2_conv_fwd_k13_vs_miopen.txt): K13 wins from 160 channels on
# both sides (ResNet stages 3 / 4: 55-58 vs 79-84 and 64-66 vs 105 us; Inception's 17x17 1x7 / 7x1: 46-55 vs 63-79 us; its
# 8x8 blocks: 23-58 vs 31-87 us), ties at 128 channels (93 vs 94 us, but saves the statistics pass) and loses on the
# 64 / 96-channel layers, whose output tile is too narrow for the 8-wave pipeline
_CONV_FWD_MIN_CH = int(os.environ.get("DTF_CONV_FWD_MIN_CH", "160"))
# BatchNorm statistics of the output in K13's epilogue (+2-4 us per layer) instead of a separate pass over it
_CONV_FWD_STATS = os.environ.get("DTF_CONV_FWD_STATS", "1") != "0"


# K16 (csrc/conv3x3_c64_kernels.hip): direct kernel of the 64 -> 64 channel 3x3 stride-1 layers (ResNet stage 1), forward
# and data gradient: 62-66 us vs MIOpen 107-112 us at batch 256, 56 x 56, and the BatchNorm statistics ride in its epilogue
_CONV_C64 = os.environ.get("DTF_CONV_C64", "1") != "0"


def conv_c64_shape(weight, stride, padding):
    """The layer shape K16 covers (tensor checks: :func:`conv_fwd_eligible`)."""
    return (_CONV_C64 and _CONV_FWD != "0" and tuple(weight.shape) == (64, 64, 3, 3) and tuple(stride) == (1, 1)
            and tuple(padding) == (1, 1))


def conv_fwd_profitable(cin, cout, weight=None, stride=None, padding=None):
    if _CONV_FWD == "0":
        return False
    if weight is not None and conv_c64_shape(weight, stride, padding):
        return True
    # deterministic mode: also the 128-channel layers, where K13 ties with MIOpen's TUNED kernel at batch 256 (93 vs 94 us)
    # but an untuned batch size gets MIOpen's heuristic pick, which accumulates with atomics -- at batch 32 ResNet-50's
    # FORWARD was not reproducible run to run because of block2.0.conv2 alone (scripts/fwd_repro_check.py)
    return _CONV_FWD == "all" or min(cin, cout) >= (min(128, _CONV_FWD_MIN_CH) if _DETERMINISTIC else _CONV_FWD_MIN_CH)


def _fwd_lib():
    global _fwd_declared
    import ctypes

    from . import hip

    lib = hip.lib()
    if not _fwd_declared:
        P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        hip.declare("dtf_conv_fwd_stats_blocks", [L, I], L)
        hip.declare("dtf_conv_fwd_set_tile", [I, I], None)
        hip.declare("dtf_conv_fwd", [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P])
        hip.declare("dtf_conv3x3_c64_ok", [I, I, I], I)
        hip.declare("dtf_conv3x3_c64_blocks", [I, I, I], L)
        hip.declare("dtf_conv3x3_c64", [P, P, P, P, P, I, I, I, P])
        _fwd_declared = True
    return lib


def conv_fwd_eligible(x, weight, stride, padding, dilation=(1, 1), groups=1):
    return (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.dim() == 4 and groups == 1
            and tuple(dilation) == (1, 1) and stride[0] == stride[1] and x.shape[1] % 32 == 0 and weight.shape[0] % 8 == 0
            and weight.shape[2] * weight.shape[3] <= 32 and x.is_contiguous(memory_format=torch.channels_last)
            and weight.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0
            and weight.data_ptr() % 16 == 0 and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 31) - 256)


def conv_fwd(x, weight, stride, padding, stats=False, _traced=False, _product="fwd"):
    """``conv2d(x, weight, None, stride, padding)`` for NHWC bf16 tensors on K13.  ``stats=True`` -> ``(y, partials)``
    with the fp32 ``[blocks, 2, Cout]`` column sums / sums of squares of ``y`` from the epilogue (what
    ``FusedBatchNormAct(..., stats=partials)`` finalizes instead of reading ``y`` once more)."""
    from . import hip

    lib = _fwd_lib()
    NB, Cin, H, W = x.shape
    Cout, _, KH, KW = weight.shape
    s = int(stride[0])
    OH = (H + 2 * int(padding[0]) - KH) // s + 1
    OW = (W + 2 * int(padding[1]) - KW) // s + 1
    dev = x.device
    z = _zero_page.get(dev)
    if z is None:
        z = torch.zeros(64, dtype=torch.bfloat16, device=dev)
        _zero_page[dev] = z
    y = torch.empty((NB, OH, OW, Cout), dtype=torch.bfloat16, device=dev)
    part = None
    if conv_c64_shape(weight, stride, padding) and lib.dtf_conv3x3_c64_ok(NB, H, W):
        if not _traced:
            _rt(_product, weight, x, stride, "K16 direct 64-channel 3x3" + (" + BN statistics" if stats else ""))
        if stats:
            part = torch.empty(int(lib.dtf_conv3x3_c64_blocks(NB, H, W)), 2, Cout, dtype=torch.float32, device=dev)
        rc = lib.dtf_conv3x3_c64(hip.ptr(x), hip.ptr(weight), hip.ptr(z), hip.ptr(y), hip.ptr(part) if stats else None, NB, H, W,
                                 hip.stream())
        hip.check(rc, "dtf_conv3x3_c64")
        y = y.permute(0, 3, 1, 2)
        return (y, part) if stats else y
    if not _traced:
        _rt(_product, weight, x, stride, "K13 implicit GEMM" + (" + BN statistics" if stats else ""))
    if stats:
        part = torch.empty(int(lib.dtf_conv_fwd_stats_blocks(NB * OH * OW, Cout)), 2, Cout, dtype=torch.float32, device=dev)
    rc = lib.dtf_conv_fwd(hip.ptr(x), hip.ptr(weight), hip.ptr(z), hip.ptr(y), hip.ptr(part) if stats else None, NB, H, W, Cin,
                          OH, OW, Cout, KH, KW, s, int(padding[0]), int(padding[1]), hip.stream())
    hip.check(rc, "dtf_conv_fwd")
    y = y.permute(0, 3, 1, 2)
    return (y, part) if stats else y


# Data gradient of a stride-1 convolution as a FORWARD convolution of dY with the spatially flipped, channel-
# transposed filter (padding K-1-p).  MIOpen's tuned forward kernels run at 500-830 TFLOP/s on the 3x3 / 1x7 / 7x1
# shapes of ResNet and Inception, its backward-data kernels at 300-490 (scripts/conv_wrw_bench.py with
# DTF_BENCH_FWD_DGRAD=1); the filter transform is two small copies of a <= 5 MB tensor.
# "all" (default): every stride-1 KxK layer -- the transposed forward problems of the shipped models are in the
# shipped tuning DB (Inception-v3 batch 256: 6 794 -> 6 920 images/s over "same"; untuned they LOSE: 6 320);
# "same": only where the transposed problem IS the layer's own forward problem (Cin == Cout, 'same' padding);
# "0": off.
_DGRAD_AS_FWD = os.environ.get("DTF_CONV_DGRAD_FWD", "all")


_ft_declared = False
_FT_KERNEL = os.environ.get("DTF_FILTER_KERNEL", "1") != "0"      # A/B switch: 0 = framework flip + copy


def _dgrad_as_fwd_ok(x, weight, stride, padding):
    if _DGRAD_AS_FWD == "0" or stride[0] != 1 or stride[1] != 1:
        return False
    kh, kw = weight.shape[2], weight.shape[3]
    if kh == 1 and kw == 1:
        return False
    if padding[0] > kh - 1 or padding[1] > kw - 1:
        return False
    if _DGRAD_AS_FWD == "all":
        return True
    return weight.shape[0] == weight.shape[1] and 2 * padding[0] == kh - 1 and 2 * padding[1] == kw - 1


def dgrad_as_forward(dy, weight, padding, force_own=False):
    """``dX`` of a stride-1, dilation-1 ``conv2d(x, weight, padding=padding)``.  ``force_own``: K13 wherever it is
    eligible, also where a tuned MIOpen kernel would be faster (strict deterministic mode)."""
    cout, cin, kh, kw = weight.shape
    if (_FT_KERNEL and weight.is_cuda and weight.dtype == torch.bfloat16
            and weight.is_contiguous(memory_format=torch.channels_last)):
        global _ft_declared
        import ctypes

        from . import hip

        lib = hip.lib()
        if not _ft_declared:
            hip.declare("dtf_filter_flip_transpose", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_void_p])
            _ft_declared = True
        buf = torch.empty((cin, kh, kw, cout), dtype=weight.dtype, device=weight.device)
        hip.check(lib.dtf_filter_flip_transpose(hip.ptr(weight), hip.ptr(buf), cout, cin, kh * kw, hip.stream()),
                  "dtf_filter_flip_transpose")
        wt = buf.permute(0, 3, 1, 2)          # [Cin, Cout, KH, KW], channels_last-dense
    else:
        wt = weight.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
    pad = (kh - 1 - padding[0], kw - 1 - padding[1])
    if (force_own or conv_fwd_profitable(cout, cin, wt, (1, 1), pad)) and conv_fwd_eligible(dy, wt, (1, 1), pad):
        return conv_fwd(dy, wt, (1, 1), pad, _product="dgrad (as forward, flipped filter)")
    _rt("dgrad (as forward, flipped filter)", wt, dy, (1, 1), "MIOpen forward kernel")
    return F.conv2d(dy, wt, None, 1, pad)


class _ConvFn(torch.autograd.Function):
    """``conv2d`` with per-product kernel choice in backward: weight gradient on K6T where that wins, data gradient
    as a forward convolution where that wins, MIOpen's backward kernels otherwise."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, stats=False):
        """``stats=True`` -> ``(y, partials)``: the BatchNorm statistics partials of ``y`` from K13's epilogue (``None``
        when the layer runs in MIOpen)."""
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, padding)
        ctx.set_materialize_grads(False)          # no zero-filled "gradient" of the statistics output
        if (conv_fwd_profitable(weight.shape[1], weight.shape[0], weight, stride, padding)
                and conv_fwd_eligible(x, weight, stride, padding)):
            if stats:
                y, part = conv_fwd(x, weight, stride, padding, stats=True)
                ctx.mark_non_differentiable(part)
                return y, part
            return conv_fwd(x, weight, stride, padding)
        _rt("fwd", weight, x, stride, "MIOpen")
        y = F.conv2d(x, weight, None, stride, padding)
        return (y, None) if stats else y

    @staticmethod
    def backward(ctx, dy, _dpart=None):
        x, weight = ctx.saved_tensors
        stride, padding = ctx.geom
        if dy is None:
            return None, None, None, None, None
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        use_tn = (ctx.needs_input_grad[1] and _CONV_WRW and conv_wrw_profitable(weight, x, stride, padding)
                  and conv_wrw_eligible(dy, x, weight, stride, padding))
        use_fwd = ctx.needs_input_grad[0] and _dgrad_as_fwd_ok(x, weight, stride, padding)
        use_img = (_DETERMINISTIC and ctx.needs_input_grad[1] and not use_tn and weight.shape[1] < 8
                   and image_wrw_eligible(dy, x, weight, stride, padding))
        if (_DETERMINISTIC_STRICT and ctx.needs_input_grad[1] and not use_tn and not use_img and _CONV_WRW and weight.shape[1] < 8
                and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16):
            # an image-input layer (3 channels; Inception-v3's 3x3 / 2 stem): no own kernel takes fewer than 8 input
            # channels and MIOpen's weight gradient accumulates with atomics -- the one tensor of Inception-v3 that
            # differed run to run.  DTF_DETERMINISTIC=2 pads the input to 8 zero-filled channels (one extra pass over
            # the image) and lets K6T-conv compute the 8-channel gradient; the first `cin` channels are the answer.
            # Not the default: a 72-wide K in a 256-wide tile is slow (Inception-v3 28.5 vs 27.5 ms/step, -3.7 %).
            cin = weight.shape[1]
            xp = torch.empty((x.shape[0], 8, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device,
                             memory_format=torch.channels_last).zero_()
            wp = torch.empty((weight.shape[0], 8, weight.shape[2], weight.shape[3]), dtype=weight.dtype,
                             device=weight.device, memory_format=torch.channels_last)
            if conv_wrw_eligible(dy, xp, wp, stride, padding, min_k=0):
                _rt("wgrad", weight, x, stride, "K6T-conv over the input padded to 8 channels")

                def padded_wrw():
                    xp[:, :cin].copy_(x)
                    return conv_wrw(dy, xp, wp, stride, padding)[:, :cin].contiguous(memory_format=torch.channels_last)

                dw = _streams.run_wgrad(padded_wrw, weight, dy, x, xp)
                if ctx.needs_input_grad[0] and not use_fwd:
                    dx, _, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, list(stride), list(padding),
                                                                   [1, 1], False, [0, 0], 1, [True, False, False])
                elif use_fwd:
                    dx = dgrad_as_forward(dy, weight, padding)
                return dx, dw, None, None, None
        strided_dx = None
        if (_DETERMINISTIC_STRICT and ctx.needs_input_grad[0] and not use_fwd and stride[0] == stride[1] and stride[0] > 1
                and x.is_cuda and dy.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
                and padding[0] <= weight.shape[2] - 1 and padding[1] <= weight.shape[3] - 1
                and weight.shape[0] % 32 == 0 and weight.shape[1] % 8 == 0):          # what K13 takes as (Cin, Cout) of the flipped filter
            # strict mode: the data gradient of a strided layer (ResNet's three 3x3 / 2 layers) as the stride-1 data gradient
            # of dY spread out with zeros between its pixels -- s^2 times the arithmetic of the library's strided kernel
            # plus one pass to build the spread tensor, but on K13 (fixed summation order) at every batch size, where
            # MIOpen's pick for a shape outside its tuning database may accumulate with atomics
            sd = int(stride[0])
            N, _, H, W = x.shape
            kh, kw = weight.shape[2], weight.shape[3]
            hu, wu = H + 2 * int(padding[0]) - kh + 1, W + 2 * int(padding[1]) - kw + 1
            OH, OW = dy.shape[2], dy.shape[3]
            up = torch.empty((N, dy.shape[1], hu, wu), dtype=dy.dtype, device=dy.device, memory_format=torch.channels_last).zero_()
            up[:, :, 0:sd * (OH - 1) + 1:sd, 0:sd * (OW - 1) + 1:sd].copy_(dy)
            strided_dx = dgrad_as_forward(up, weight, padding, force_own=True)
        mask = [ctx.needs_input_grad[0] and not use_fwd and strided_dx is None,
                ctx.needs_input_grad[1] and not use_tn and not use_img, False]
        if use_img:
            _rt("wgrad", weight, x, stride, "K21 patch matrix + K6T")
        if mask[0]:
            _rt("dgrad", weight, x, stride, "MIOpen")
        if mask[1]:
            _rt("wgrad", weight, x, stride, "MIOpen")
        if use_tn:
            _rt("wgrad", weight, x, stride, "K14 direct" if conv_wrw_direct_ok(x, weight, stride, padding) else "K6T-conv")
        if mask[0] and mask[1] and _streams.ENABLED:
            # the library's data gradient on this stream, its weight gradient on the side stream
            dx, _, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, list(stride), list(padding), [1, 1], False,
                                                           [0, 0], 1, [True, False, False])
            dw = _streams.run_wgrad(lambda: torch.ops.aten.convolution_backward(
                dy, x, weight, None, list(stride), list(padding), [1, 1], False, [0, 0], 1, [False, True, False])[1], weight, dy, x)
        elif mask[0] or mask[1]:
            if mask[1] and not mask[0]:
                dw = _streams.run_wgrad(lambda: torch.ops.aten.convolution_backward(
                    dy, x, weight, None, list(stride), list(padding), [1, 1], False, [0, 0], 1, [False, True, False])[1], weight, dy, x)
            else:
                gx, gw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, list(stride), list(padding), [1, 1],
                                                                False, [0, 0], 1, mask)
                dx, dw = (gx if mask[0] else None), (gw if mask[1] else None)
        if use_fwd:
            dx = dgrad_as_forward(dy, weight, padding, force_own=_DETERMINISTIC_STRICT)
        if strided_dx is not None:
            dx = strided_dx
        if use_tn:
            dw = _streams.run_wgrad(lambda: conv_wrw(dy, x, weight, stride, padding), weight, dy, x)
        if use_img:
            dw = _streams.run_wgrad(lambda: image_wrw(dy, x, weight, stride, padding), weight, dy, x)
        return dx, dw, None, None, None


class Conv2dTN(nn.Conv2d):
    """``nn.Conv2d`` (bias-free, groups 1) whose backward picks the kernel per product (see :class:`_ConvFn`)."""

    def _own(self, x):
        """Route through :class:`_ConvFn`?  -> (yes / no, forward on K13)."""
        if not (self.bias is None and self.groups == 1 and x.is_cuda and x.dtype == torch.bfloat16
                and isinstance(self.padding, tuple) and self.dilation == (1, 1)):
            return False, False
        k13 = (conv_fwd_profitable(self.in_channels, self.out_channels, self.weight, self.stride, self.padding)
               and conv_fwd_eligible(x, self.weight, self.stride, self.padding))
        bwd = torch.is_grad_enabled() and ((_CONV_WRW and conv_wrw_profitable(self.weight, x, self.stride, self.padding))
                                           or _dgrad_as_fwd_ok(x, self.weight, self.stride, self.padding))
        return k13 or bwd, k13

    def forward(self, x):
        if self._own(x)[0]:
            return _ConvFn.apply(x, self.weight, self.stride, self.padding, False)
        _rt("fwd + bwd", self.weight, x, self.stride, "framework conv2d (MIOpen)")
        return super().forward(x)

    def forward_stats(self, x):
        """-> ``(self(x), stats or None)`` for a following ``FusedBatchNormAct(y, ..., stats=stats)`` in training."""
        if _CONV_FWD_STATS and self.training and self._own(x)[1]:
            return _ConvFn.apply(x, self.weight, self.stride, self.padding, True)
        return self(x), None
