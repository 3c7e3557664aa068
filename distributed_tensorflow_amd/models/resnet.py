"""ResNet-50/101/152 v1 for synthetic ImageNet (config #2; the headline
``tf_cnn_benchmarks --model=resnet50 --variable_update=parameter_server``
workload of BASELINE.md).  There is no torchvision in the image, so the network
is defined here.

MI355X specifics: activations are NHWC (``channels_last``) bf16 end to end;
convolutions run in MIOpen through PyTorch-ROCm; everything between two
convolutions -- BatchNorm, the residual add and the ReLU -- is one
:class:`~distributed_tensorflow_amd.ops.batchnorm.FusedBatchNormAct` layer
(hand-written gfx950 kernels, K7).  ``version="v1.5"`` (default) puts the
stride on the 3x3 convolution (the MLPerf / torchvision variant, 4.1 GMACs);
``"v1"`` puts it on the first 1x1 like the original (3.8 GMACs).

Variables are named like TF-slim's ``resnet_v1_50`` for checkpoints
(``tf_variable_names``): ``resnet_v1_50/block2/unit_3/bottleneck_v1/conv2/weights``,
``.../BatchNorm/gamma|beta|moving_mean|moving_variance``, ``resnet_v1_50/logits/...``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops.batchnorm import FusedBatchNormAct, fused_bn2_add_relu
from ..ops.conv import Conv1x1, Conv2dTN, conv1x1, k17_ok, subsample_fork
from ..ops.pool import global_avg_pool
from ..ops.stem import stem_conv_bn_relu_pool

_SUBSAMPLE = __import__("os").environ.get("DTF_RESNET_SUBSAMPLE", "1") != "0"      # A/B switch
_DEPTHS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _conv(i, o, k, stride=1):
    if k == 1 and stride == 1:
        return Conv1x1(i, o)          # NHWC 1x1 = GEMM: per-product kernel choice (ops/conv.py)
    return Conv2dTN(i, o, k, stride=stride, padding=k // 2, bias=False)   # wide layers: K6T weight gradient


def _conv_pre(conv, x):
    """-> ``(conv(x), statistics partials or None)``: the 64 -> 256 products of the first stage run on K17, whose
    epilogue already holds the BatchNorm statistics."""
    if isinstance(conv, Conv1x1) and k17_ok(x, conv.weight):
        return conv.forward_stats(x)
    return conv(x), None


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, v15):
        super().__init__()
        out = planes * self.expansion
        self.conv1 = _conv(inplanes, planes, 1, 1 if v15 else stride)
        self.bn1 = FusedBatchNormAct(planes, relu=True)
        self.conv2 = _conv(planes, planes, 3, stride if v15 else 1)
        self.bn2 = FusedBatchNormAct(planes, relu=True)
        self.conv3 = _conv(planes, out, 1)
        self.bn3 = FusedBatchNormAct(out, relu=True)          # fused: BN + shortcut add + ReLU
        if stride != 1 or inplanes != out:
            self.shortcut_conv = _conv(inplanes, out, 1, stride)
            self.shortcut_bn = FusedBatchNormAct(out, relu=False)
        else:
            self.shortcut_conv = None

    def _conv2_bn2(self, h):
        if isinstance(self.conv2, Conv2dTN):
            c, st = self.conv2.forward_stats(h)      # wide layers: K13 implicit GEMM, BN statistics from its epilogue
            return self.bn2(c, stats=st)
        return self.bn2(self.conv2(h))

    def forward(self, x):
        sc_stride = self.shortcut_conv.stride[0] if self.shortcut_conv is not None else 1
        if (_SUBSAMPLE and sc_stride > 1 and self.shortcut_conv.kernel_size == (1, 1) and isinstance(self.conv1, Conv1x1)
                and self.training and x.is_cuda and x.dtype == torch.bfloat16 and x.requires_grad
                and x.is_contiguous(memory_format=torch.channels_last)):
            # strided 1x1 shortcut = dense 1x1 GEMM of the subsampled input; its gradient is added on those pixels only
            xs, x = subsample_fork(x, sc_stride)
            s = conv1x1(xs, self.shortcut_conv.weight, mm_fwd=True)
            y = self._conv2_bn2(self.bn1(self.conv1(x)))
            return fused_bn2_add_relu(self.conv3(y), self.bn3, s, self.shortcut_bn)
        if isinstance(self.conv1, Conv1x1):
            # x feeds conv1 and the shortcut: the fork sums their two gradients inside conv1's dX GEMM.
            # The shortcut's gradient comes from the K7 residual epilogue / the shortcut conv's dgrad:
            # a fresh tensor nobody else reads, so the GEMM may accumulate into it in place.
            y, x = self.conv1.forward_fork(x, own_skip_grad=True)
        else:
            y = self.conv1(x)
        if self.shortcut_conv is not None and self.training and x.is_cuda:
            # both BatchNorms, the add and the ReLU of a stage's first block in one K7 node
            s, st_s = _conv_pre(self.shortcut_conv, x)
            y = self._conv2_bn2(self.bn1(y))
            y, st_y = _conv_pre(self.conv3, y)
            return fused_bn2_add_relu(y, self.bn3, s, self.shortcut_bn, st_y, st_s)
        if self.shortcut_conv is None:
            sc = x
        elif isinstance(self.shortcut_conv, Conv1x1):
            s, st = self.shortcut_conv.forward_stats(x)      # BN statistics from the GEMM epilogue (K6 STATS)
            sc = self.shortcut_bn(s, stats=st)
        else:
            sc = self.shortcut_bn(self.shortcut_conv(x))
        y = self._conv2_bn2(self.bn1(y))
        if isinstance(self.conv3, Conv1x1):
            y, st = self.conv3.forward_stats(y)
            return self.bn3(y, sc, stats=st)
        return self.bn3(self.conv3(y), sc)


class ResNet(nn.Module):
    def __init__(self, depth=50, num_classes=1000, version="v1.5", zero_init_residual=False):
        super().__init__()
        self.depth = depth
        v15 = version in ("v1.5", "1.5", "v15")
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FusedBatchNormAct(64, relu=True)
        blocks, inplanes = [], 64
        for stage, n in enumerate(_DEPTHS[depth]):
            planes = 64 << stage
            units = []
            for u in range(n):
                units.append(Bottleneck(inplanes, planes, 2 if (u == 0 and stage > 0) else 1, v15))
                inplanes = planes * 4
            blocks.append(nn.Sequential(*units))
        self.block1, self.block2, self.block3, self.block4 = blocks
        self.fc = nn.Linear(inplanes, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.zeros_(m.bn3.weight)

    def forward(self, x):
        # K15 stem: direct 7x7 convolution with the BatchNorm statistics in its epilogue, BatchNorm + ReLU applied inside
        # the pooling kernel (the normalised 112x112 tensor is never materialised), BatchNorm backward inside the
        # weight-gradient kernel (falls back to conv -> bn_relu_max_pool2d where the kernels do not apply)
        x = stem_conv_bn_relu_pool(x, self.conv1, self.bn1, 3, 2, 1)
        x = self.block4(self.block3(self.block2(self.block1(x))))
        x = global_avg_pool(x)
        return self.fc(x)


def build(depth=50, version="v1.5", num_classes=1000, **_):
    return ResNet(int(depth), int(num_classes), version)


def prepare(model, device):
    device = torch.device(device)
    if device.type == "cuda":
        return model.to(device=device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    return model.to(device)


def tf_variable_names(model):
    scope = "resnet_v1_%d" % model.depth
    bn = {"weight": "gamma", "bias": "beta", "running_mean": "moving_mean", "running_var": "moving_variance"}
    out = {}
    names = [n for n, _ in model.named_parameters()] + [n for n, _ in model.named_buffers()]
    for n in names:
        parts = n.split(".")
        if parts[0] == "conv1":
            tf = "conv1/weights"
        elif parts[0] == "bn1":
            tf = "conv1/BatchNorm/" + bn[parts[1]]
        elif parts[0] == "fc":
            tf = "logits/" + ("weights" if parts[1] == "weight" else "biases")
        else:
            unit = "%s/unit_%d/bottleneck_v1/" % (parts[0], int(parts[1]) + 1)
            layer, leaf = parts[2], parts[3]
            if layer.startswith("shortcut"):
                tf = unit + "shortcut/" + ("weights" if layer.endswith("conv") else "BatchNorm/" + bn[leaf])
            elif layer.startswith("conv"):
                tf = unit + layer + "/weights"
            else:
                tf = unit + "conv" + layer[2:] + "/BatchNorm/" + bn[leaf]
        out[n] = scope + "/" + tf
    return out


def synthetic_batch(batch, device="cpu", dtype=None, seed=None, image_size=224, num_classes=1000, **_):
    device = torch.device(device)
    g = torch.Generator().manual_seed(seed) if seed is not None else None
    dtype = dtype or (torch.bfloat16 if device.type == "cuda" else torch.float32)
    x = torch.randn(batch, 3, image_size, image_size, generator=g).to(device=device, dtype=dtype)
    x = x.contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, num_classes, (batch,), generator=g).to(device)
    return x, y


def loss_fn(model, batch):
    x, y = batch
    return F.cross_entropy(model(x).float(), y)


def default_optimizer():
    from ..parallel.ps_runtime import OptimizerSpec

    # tf_cnn_benchmarks defaults: momentum 0.9, weight decay 4e-5 (1e-4 is the ImageNet norm)
    return OptimizerSpec(kind="momentum", lr=0.1, momentum=0.9, weight_decay=1e-4)
