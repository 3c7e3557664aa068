"""Inception-v3 for synthetic ImageNet at 299x299 (config #5: sync-vs-async PS update
with job cleanup; the ``tf_cnn_benchmarks --model=inception3`` row of BASELINE.md).

Same MI355X recipe as ResNet: NHWC bf16, MIOpen convolutions, and every
``conv -> BatchNorm -> ReLU`` unit (TF-slim's ``conv2d`` with ``batch_norm``) uses the K7
fused BN+ReLU kernels.  ~23.8 M parameters without the auxiliary head (``aux_logits=True``
adds the ~3.4 M-parameter side classifier used by the TF-slim training recipe).  The 3x3/1 pooling
branches average over real pixels only (TF's 'SAME' avg_pool; ``count_include_pad=False``).
Variable names follow TF-slim's ``InceptionV3`` scope (``InceptionV3/Conv2d_1a_3x3/weights``,
``InceptionV3/Mixed_5b/Branch_1/Conv2d_0a_1x1/BatchNorm/beta``, ``InceptionV3/Logits/...``).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops.batchnorm import FusedBatchNormAct, GradSlices, bn_act_concat
from ..ops.conv import Conv1x1, Conv2dTN, conv1x1
from ..ops.pool import avg_pool2d, bn_relu_max_pool2d, global_avg_pool, max_pool2d


class ConvBN(nn.Module):
    """slim.conv2d: conv (no bias) + BatchNorm(eps=1e-3) + ReLU."""

    def __init__(self, i, o, k, stride=1, padding=0):
        super().__init__()
        if k == 1 and stride == 1 and padding == 0:
            self.conv = Conv1x1(i, o)     # NHWC 1x1 = GEMM (ops/conv.py picks the kernel per product)
        else:
            self.conv = Conv2dTN(i, o, k, stride=stride, padding=padding, bias=False)
        self.bn = FusedBatchNormAct(o, relu=True, eps=1e-3)

    def forward(self, x):
        if isinstance(self.conv, Conv2dTN):
            y, st = self.conv.forward_stats(x)       # wide layers: K13 implicit GEMM, BN statistics from its epilogue
            return self.bn(y, stats=st)
        return self.bn(self.conv(x))

    def forward_fork(self, x):
        """-> (self(x), x'): 1x1 heads of a mixed block hand their input on to the next branch, so the
        branch gradients of ``x`` are summed inside the dX GEMMs instead of by separate add kernels."""
        if isinstance(self.conv, Conv1x1):
            y, x = self.conv.forward_fork(x, own_skip_grad=True)
            return self.bn(y), x
        return self.bn(self.conv(x)), x


def _conv_fork(m, x):
    """Convolution of ConvBN ``m`` only (its BN runs inside the block's concat) + the input fork."""
    if isinstance(m.conv, Conv1x1):
        return m.conv.forward_fork(x, own_skip_grad=True)
    return m.conv(x), x


def _branch(branch, x):
    """Run a branch up to -- not including -- its last BatchNorm: -> ((conv_out, bn), x').  The 1x1 head
    forks ``x`` so the branch gradients of the block input are summed inside the dX GEMMs; the last BN of
    every branch runs in :func:`bn_act_concat`, which writes straight into the block's output."""
    if isinstance(branch, ConvBN):
        y, x = _conv_fork(branch, x)
        return (y, branch.bn), x
    mods = list(branch)
    h, x = mods[0].forward_fork(x)
    for m in mods[1:-1]:
        h = m(h)
    return (mods[-1].conv(h), mods[-1].bn), x


def _pre(m, x):
    return m.conv(x), m.bn


_FUSE_HEADS = os.environ.get("DTF_INCEPTION_FUSE_HEADS", "1") != "0"      # A/B switch
_POOL_IN_PLACE = os.environ.get("DTF_INCEPTION_POOL_IN_PLACE", "1") != "0"  # A/B: K9 avg pool reads / writes the head slice in place
_HEADS_MM = os.environ.get("DTF_INCEPTION_HEADS_MM", "0") != "0"          # fused heads forward: GEMM library (1) / MIOpen (0)
_GRAD_IN_PLACE = os.environ.get("DTF_INCEPTION_GRAD_IN_PLACE", "1") != "0"  # A/B switch: BN dx into the shared buffer
_SLICE_VIEWS = os.environ.get("DTF_INCEPTION_SLICE_VIEWS", "1") != "0"    # A/B switch: BN reads the head slices in place


class _SplitChannels(torch.autograd.Function):
    """``y [N, sum(sizes), H, W]`` (NHWC) -> dense per-branch tensors; backward concatenates the branch gradients
    into ONE tensor (autograd's own slicing would build a zero-filled full-width gradient per branch and add them)."""

    @staticmethod
    def forward(ctx, y, dense, prov, *sizes):
        ctx.prov = prov
        # dense[i]: branch i needs its own dense tensor (its consumer is not a K7 BatchNorm, which reads the
        # channel slice of ``y`` in place through its row stride)
        ctx.sizes = sizes
        outs, a = [], 0
        for c, d in zip(sizes, dense):
            v = y[:, a:a + c]
            outs.append(v.contiguous(memory_format=torch.channels_last) if (d or a % 8 or c % 8) else v)
            a += c
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        # the K7 BatchNorm layers that read a slice in place have already written their dx into the shared buffer
        # (ops/batchnorm.py::GradSlices); everything else is copied into its slice
        prov = ctx.prov
        buf, a = prov.get(), 0
        for g, c in zip(grads, ctx.sizes):
            if g is None:
                buf[:, a:a + c].zero_()
            elif not prov.holds(g, a, c):
                buf[:, a:a + c].copy_(g)
            a += c
        prov.release()
        return (buf, None, None) + (None,) * len(ctx.sizes)


def _heads(x, heads, dense=None):
    """The 1x1 convolutions that open the branches of a mixed block all read the block input: run them as ONE
    GEMM over the concatenated filters (the input is read once instead of once per branch, the data gradient is
    one GEMM instead of a chain of read-modify-write accumulations, the weight gradient one K6T call; the filters
    stay separate parameters with their TF-slim names).  -> the branch pre-BatchNorm tensors, or None if the
    block input does not qualify (CPU / fp32 runs keep the plain per-branch graph)."""
    if not (_FUSE_HEADS and x.is_cuda and x.dtype == torch.bfloat16 and all(isinstance(m.conv, Conv1x1) for m in heads)):
        return None
    w = torch.cat([m.conv.weight for m in heads], 0).contiguous(memory_format=torch.channels_last)
    y = conv1x1(x, w, mm_fwd=_HEADS_MM)
    dense = tuple(dense) if dense is not None else (not _SLICE_VIEWS,) * len(heads)
    if not _SLICE_VIEWS:
        dense = (True,) * len(heads)
    prov = GradSlices(y.shape, y.dtype, y.device)
    outs = _SplitChannels.apply(y, dense, prov, *[m.conv.out_channels for m in heads])
    a = 0
    for o, d, m in zip(outs, dense, heads):
        if not d and _GRAD_IN_PLACE:
            o._dtf_gradslot = (prov, a, m.conv.out_channels)
        a += m.conv.out_channels
    return outs


def _tail(branch, h):
    """Rest of a Sequential branch whose head convolution output ``h`` (pre-BN) is given -> (conv_out, bn)."""
    mods = list(branch)
    h = mods[0].bn(h)
    for m in mods[1:-1]:
        h = m(h)
    return mods[-1].conv(h), mods[-1].bn


_POOL_AFTER_CONV = os.environ.get("DTF_POOL_AFTER_CONV", "1") != "0"      # A/B switch


def _pool_branch(m, x):
    """The ``AvgPool_0a_3x3 -> Conv2d_0b_1x1`` branch of a mixed block, up to its BatchNorm.  Both ops are linear
    and act on different axes (the pool per channel over pixels -- with ``count_include_pad=False`` a per-pixel
    weighting --, the 1x1 convolution per pixel over channels), so they commute exactly:
    ``conv(pool(x)) == pool(conv(x))``.  Run in the second order the pooling kernels see the branch's 32..192
    output channels instead of the block's 192..2048 input channels (4-10x less HBM traffic, forward and
    backward); same parameters, same function up to bf16 rounding of the intermediate."""
    if _POOL_AFTER_CONV:
        return avg_pool2d(m.conv(x), 3, 1, 1, count_include_pad=False), m.bn
    return m.conv(avg_pool2d(x, 3, 1, 1, count_include_pad=False)), m.bn


class InceptionA(nn.Module):          # Mixed_5b/5c/5d
    def __init__(self, i, pool):
        super().__init__()
        self.b0 = ConvBN(i, 64, 1)
        self.b1 = nn.Sequential(ConvBN(i, 48, 1), ConvBN(48, 64, 5, padding=2))
        self.b2 = nn.Sequential(ConvBN(i, 64, 1), ConvBN(64, 96, 3, padding=1), ConvBN(96, 96, 3, padding=1))
        self.b3 = ConvBN(i, pool, 1)

    def forward(self, x):
        hs = _heads(x, [self.b0, self.b1[0], self.b2[0], self.b3], dense=(False, False, False, not _POOL_IN_PLACE)) if _POOL_AFTER_CONV else None
        if hs is not None:
            pool = avg_pool2d(hs[3], 3, 1, 1, count_include_pad=False)
            return bn_act_concat([(hs[0], self.b0.bn), _tail(self.b1, hs[1]), _tail(self.b2, hs[2]), (pool, self.b3.bn)])
        p0, x = _branch(self.b0, x)
        p1, x = _branch(self.b1, x)
        p2, x = _branch(self.b2, x)
        return bn_act_concat([p0, p1, p2, _pool_branch(self.b3, x)])


class InceptionB(nn.Module):          # Mixed_6a (grid reduction)
    def __init__(self, i):
        super().__init__()
        self.b0 = ConvBN(i, 384, 3, stride=2)
        self.b1 = nn.Sequential(ConvBN(i, 64, 1), ConvBN(64, 96, 3, padding=1), ConvBN(96, 96, 3, stride=2))

    def forward(self, x):
        p1, x = _branch(self.b1, x)
        return bn_act_concat([_pre(self.b0, x), p1], extra=[max_pool2d(x, 3, 2)])


class InceptionC(nn.Module):          # Mixed_6b..6e
    def __init__(self, i, c7):
        super().__init__()
        self.b0 = ConvBN(i, 192, 1)
        self.b1 = nn.Sequential(ConvBN(i, c7, 1), ConvBN(c7, c7, (1, 7), padding=(0, 3)),
                                ConvBN(c7, 192, (7, 1), padding=(3, 0)))
        self.b2 = nn.Sequential(ConvBN(i, c7, 1), ConvBN(c7, c7, (7, 1), padding=(3, 0)),
                                ConvBN(c7, c7, (1, 7), padding=(0, 3)), ConvBN(c7, c7, (7, 1), padding=(3, 0)),
                                ConvBN(c7, 192, (1, 7), padding=(0, 3)))
        self.b3 = ConvBN(i, 192, 1)

    def forward(self, x):
        hs = _heads(x, [self.b0, self.b1[0], self.b2[0], self.b3], dense=(False, False, False, not _POOL_IN_PLACE)) if _POOL_AFTER_CONV else None
        if hs is not None:
            pool = avg_pool2d(hs[3], 3, 1, 1, count_include_pad=False)
            return bn_act_concat([(hs[0], self.b0.bn), _tail(self.b1, hs[1]), _tail(self.b2, hs[2]), (pool, self.b3.bn)])
        p0, x = _branch(self.b0, x)
        p1, x = _branch(self.b1, x)
        p2, x = _branch(self.b2, x)
        return bn_act_concat([p0, p1, p2, _pool_branch(self.b3, x)])


class InceptionD(nn.Module):          # Mixed_7a (grid reduction)
    def __init__(self, i):
        super().__init__()
        self.b0 = nn.Sequential(ConvBN(i, 192, 1), ConvBN(192, 320, 3, stride=2))
        self.b1 = nn.Sequential(ConvBN(i, 192, 1), ConvBN(192, 192, (1, 7), padding=(0, 3)),
                                ConvBN(192, 192, (7, 1), padding=(3, 0)), ConvBN(192, 192, 3, stride=2))

    def forward(self, x):
        p0, x = _branch(self.b0, x)
        p1, x = _branch(self.b1, x)
        return bn_act_concat([p0, p1], extra=[max_pool2d(x, 3, 2)])


class InceptionE(nn.Module):          # Mixed_7b/7c
    def __init__(self, i):
        super().__init__()
        self.b0 = ConvBN(i, 320, 1)
        self.b1 = ConvBN(i, 384, 1)
        self.b1a = ConvBN(384, 384, (1, 3), padding=(0, 1))
        self.b1b = ConvBN(384, 384, (3, 1), padding=(1, 0))
        self.b2 = nn.Sequential(ConvBN(i, 448, 1), ConvBN(448, 384, 3, padding=1))
        self.b2a = ConvBN(384, 384, (1, 3), padding=(0, 1))
        self.b2b = ConvBN(384, 384, (3, 1), padding=(1, 0))
        self.b3 = ConvBN(i, 192, 1)

    def forward(self, x):
        hs = _heads(x, [self.b0, self.b1, self.b2[0], self.b3], dense=(False, False, False, not _POOL_IN_PLACE)) if _POOL_AFTER_CONV else None
        if hs is not None:
            y1 = self.b1.bn(hs[1])
            y2 = self.b2[1](self.b2[0].bn(hs[2]))
            pool = avg_pool2d(hs[3], 3, 1, 1, count_include_pad=False)
            return bn_act_concat([(hs[0], self.b0.bn), _pre(self.b1a, y1), _pre(self.b1b, y1), _pre(self.b2a, y2),
                                  _pre(self.b2b, y2), (pool, self.b3.bn)])
        p0, x = _branch(self.b0, x)
        y1, x = self.b1.forward_fork(x)
        h2, x = self.b2[0].forward_fork(x)
        y2 = self.b2[1](h2)
        return bn_act_concat([p0, _pre(self.b1a, y1), _pre(self.b1b, y1), _pre(self.b2a, y2), _pre(self.b2b, y2),
                              _pool_branch(self.b3, x)])


class AuxHead(nn.Module):
    def __init__(self, i, num_classes):
        super().__init__()
        self.conv0 = ConvBN(i, 128, 1)
        self.conv1 = ConvBN(128, 768, 5)
        self.fc = nn.Linear(768, num_classes)

    def forward(self, x):
        x = avg_pool2d(x, 5, 3)
        x = self.conv1(self.conv0(x))
        return self.fc(x.mean((2, 3)))


class InceptionV3(nn.Module):
    def __init__(self, num_classes=1000, aux_logits=False):
        super().__init__()
        self.Conv2d_1a_3x3 = ConvBN(3, 32, 3, stride=2)
        self.Conv2d_2a_3x3 = ConvBN(32, 32, 3)
        self.Conv2d_2b_3x3 = ConvBN(32, 64, 3, padding=1)
        self.Conv2d_3b_1x1 = ConvBN(64, 80, 1)
        self.Conv2d_4a_3x3 = ConvBN(80, 192, 3)
        self.Mixed_5b = InceptionA(192, 32)
        self.Mixed_5c = InceptionA(256, 64)
        self.Mixed_5d = InceptionA(288, 64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, 128)
        self.Mixed_6c = InceptionC(768, 160)
        self.Mixed_6d = InceptionC(768, 160)
        self.Mixed_6e = InceptionC(768, 192)
        self.AuxLogits = AuxHead(768, num_classes) if aux_logits else None
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.Logits = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.1, a=-0.2, b=0.2)

    def forward(self, x):
        # the BatchNorm + ReLU in front of each stem pool runs inside the pooling kernel (ops/pool.py)
        x = self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x))
        x = bn_relu_max_pool2d(self.Conv2d_2b_3x3.conv(x), self.Conv2d_2b_3x3.bn, 3, 2)
        x = self.Conv2d_3b_1x1(x)
        x = bn_relu_max_pool2d(self.Conv2d_4a_3x3.conv(x), self.Conv2d_4a_3x3.bn, 3, 2)
        x = self.Mixed_5d(self.Mixed_5c(self.Mixed_5b(x)))
        x = self.Mixed_6e(self.Mixed_6d(self.Mixed_6c(self.Mixed_6b(self.Mixed_6a(x)))))
        aux = self.AuxLogits(x) if (self.AuxLogits is not None and self.training) else None
        x = self.Mixed_7c(self.Mixed_7b(self.Mixed_7a(x)))
        logits = self.Logits(global_avg_pool(x))
        return (logits, aux) if aux is not None else logits


def build(num_classes=1000, aux_logits=False, **_):
    if isinstance(aux_logits, str):
        aux_logits = aux_logits.lower() in ("1", "true", "yes")
    return InceptionV3(int(num_classes), bool(aux_logits))


def prepare(model, device):
    device = torch.device(device)
    if device.type == "cuda":
        return model.to(device=device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    return model.to(device)


def tf_variable_names(model):
    bn = {"weight": "gamma", "bias": "beta", "running_mean": "moving_mean", "running_var": "moving_variance"}
    out = {}
    names = [n for n, _ in model.named_parameters()] + [n for n, _ in model.named_buffers()]
    for n in names:
        parts = n.split(".")
        leaf = parts[-1]
        if parts[0] == "Logits":
            tf = "Logits/Conv2d_1c_1x1/" + ("weights" if leaf == "weight" else "biases")
        elif parts[0] == "AuxLogits" and parts[1] == "fc":
            tf = "AuxLogits/Conv2d_2b_1x1/" + ("weights" if leaf == "weight" else "biases")
        else:
            scope = "/".join(p for p in parts[:-2])
            # branch modules b0..b3 -> Branch_0..3 ; sequential indices -> Conv2d_0<letter>
            toks = []
            for p in parts[:-2]:
                if p[0] == "b" and p[1:2].isdigit():
                    toks.append("Branch_%s%s" % (p[1], ("_" + p[2:]) if len(p) > 2 else ""))
                elif p.isdigit():
                    toks.append("Conv2d_0%s" % "abcdefgh"[int(p)])
                else:
                    toks.append(p)
            scope = "/".join(toks)
            tf = scope + ("/weights" if parts[-2] == "conv" else "/BatchNorm/" + bn[leaf])
        out[n] = "InceptionV3/" + tf
    return out


def synthetic_batch(batch, device="cpu", dtype=None, seed=None, image_size=299, num_classes=1000, **_):
    device = torch.device(device)
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    dtype = dtype or (torch.bfloat16 if device.type == "cuda" else torch.float32)
    x = torch.randn(batch, 3, image_size, image_size, generator=g).to(device=device, dtype=dtype)
    x = x.contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, num_classes, (batch,), generator=g).to(device)
    return x, y


def loss_fn(model, batch):
    x, y = batch
    out = model(x)
    if isinstance(out, tuple):
        return F.cross_entropy(out[0].float(), y) + 0.4 * F.cross_entropy(out[1].float(), y)
    return F.cross_entropy(out.float(), y)


def default_optimizer():
    from ..parallel.ps_runtime import OptimizerSpec

    # tf_cnn_benchmarks trains every model with momentum SGD by default
    return OptimizerSpec(kind="momentum", lr=0.045, momentum=0.9, weight_decay=4e-5)
