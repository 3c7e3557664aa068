"""K15 (the ResNet stem kernels, ops/csrc/stem_kernels.hip) against plain fp32 PyTorch references."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _inputs(dev, N, H, W, seed=0):
    torch.manual_seed(seed)
    x = torch.randn(N, 3, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, device=dev) * 147 ** -0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return x, w


@pytest.mark.parametrize("cfg", [(2, 224, 224), (3, 64, 72), (1, 37, 88), (2, 7, 64), (5, 96, 104)])
def test_stem_forward_matches_reference(dev, cfg):
    """Direct 7x7 / 2 convolution + statistics epilogue vs fp32 conv2d: full-size images, odd heights, column fragments
    and row strips that end inside the tile, the smallest image."""
    from distributed_tensorflow_amd.ops import stem

    N, H, W = cfg
    x, w = _inputs(dev, N, H, W)
    assert stem.eligible(x, w)
    ref = F.conv2d(x.float(), w.float(), None, 2, 3)
    y, part = stem.stem_conv_fwd(x, w)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=2e-2)
    assert float((y.float() - ref).norm() / ref.norm()) < 5e-3
    yf = y.float()
    torch.testing.assert_close(part[:, 0].sum(0), yf.sum((0, 2, 3)), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(part[:, 1].sum(0), (yf * yf).sum((0, 2, 3)), rtol=1e-4, atol=2e-2)


class _Ref(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn = nn.BatchNorm2d(64)

    def forward(self, x):
        return F.max_pool2d(F.relu(self.bn(self.conv(x))), 3, 2, 1)


@pytest.mark.parametrize("cfg", [(4, 64, 64), (2, 224, 224), (3, 40, 72)])
@pytest.mark.parametrize("fused_wrw", [True, False])
def test_stem_node_matches_fp32_composition(dev, cfg, fused_wrw, monkeypatch):
    """The whole stem node (conv + statistics, finalize, BN/ReLU inside the pool; pool gather, BN backward reduction, weight
    gradient with the BN backward applied on load) vs conv2d -> batch_norm -> relu -> max_pool2d in fp32: output, running
    statistics, gradients of the filter, gamma and beta."""
    from distributed_tensorflow_amd.ops import batchnorm as bnmod, stem

    if fused_wrw and not hasattr(stem._lib(), "dtf_stem_wrw"):
        pytest.skip("fused weight gradient not built")
    monkeypatch.setattr(stem, "_WRW", fused_wrw)
    N, H, W = cfg
    x, w = _inputs(dev, N, H, W, seed=1)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    bn = bnmod.FusedBatchNormAct(64, relu=True).to(dev).to(torch.bfloat16).train()
    with torch.no_grad():
        conv.weight.copy_(w)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref = _Ref().to(dev).float().train()
    with torch.no_grad():
        ref.conv.weight.copy_(conv.weight.float())
        ref.bn.weight.copy_(bn.weight.float())
        ref.bn.bias.copy_(bn.bias.float())
    assert stem.eligible(x, conv.weight)
    y = stem.stem_conv_bn_relu_pool(x, conv, bn, 3, 2, 1)
    assert y.grad_fn is not None and "Stem" in type(y.grad_fn).__name__
    yr = ref(x.float())
    torch.testing.assert_close(y.float(), yr, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(bn.running_mean.float(), ref.bn.running_mean, rtol=1e-2, atol=1e-3)
    torch.testing.assert_close(bn.running_var.float(), ref.bn.running_var, rtol=1e-2, atol=1e-3)
    g = torch.randn_like(yr)
    y.backward(g.to(torch.bfloat16))
    yr.backward(g.to(torch.bfloat16).float())

    def rel(a, b):
        return float((a.float() - b).norm() / b.norm())

    # the filter gradient sees the bf16 rounding of the convolution output through the ReLU mask and the pool's winning
    # taps (the library path is as far from fp32 as the fused one); the two paths are compared with each other below
    assert rel(conv.weight.grad, ref.conv.weight.grad) < 1e-1
    assert rel(bn.weight.grad, ref.bn.weight.grad) < 3e-2
    assert rel(bn.bias.grad, ref.bn.bias.grad) < 3e-2
    assert conv.weight.grad.is_contiguous(memory_format=torch.channels_last)


def test_stem_fused_weight_gradient_matches_unfused(dev, monkeypatch):
    """BatchNorm backward applied inside the weight-gradient kernel (fp32 in registers) vs the K7 apply pass (bf16 tensor)
    + library weight gradient: same operands, so the two agree to bf16 rounding of that one tensor."""
    from distributed_tensorflow_amd.ops import batchnorm as bnmod, stem

    if not hasattr(stem._lib(), "dtf_stem_wrw"):
        pytest.skip("fused weight gradient not built")
    x, w = _inputs(dev, 6, 96, 80, seed=2)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    bn = bnmod.FusedBatchNormAct(64, relu=True).to(dev).to(torch.bfloat16).train()
    grads = []
    for fused in (True, False):
        monkeypatch.setattr(stem, "_WRW", fused)
        conv.weight.grad = None
        y = stem.stem_conv_bn_relu_pool(x, conv, bn, 3, 2, 1)
        torch.manual_seed(5)
        y.backward(torch.randn_like(y))
        grads.append(conv.weight.grad.float().clone())
    assert float((grads[0] - grads[1]).norm() / grads[1].norm()) < 1.5e-2


# ---------------------------------------------------------------------------------------------------------------------
# K16: direct 64 -> 64 channel 3x3 convolution (ops/csrc/conv3x3_c64_kernels.hip)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [(2, 56, 56), (3, 17, 23), (1, 16, 16), (2, 9, 40), (5, 28, 28), (1, 3, 62)])
def test_conv3x3_c64_matches_reference(dev, cfg):
    """Forward + statistics epilogue vs fp32 conv2d: fragments that wrap around a row end, strips that end inside the
    image, odd widths, the smallest width."""
    from distributed_tensorflow_amd.ops import conv

    N, H, W = cfg
    torch.manual_seed(0)
    x = torch.randn(N, 64, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, device=dev) * 576 ** -0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert conv.conv_c64_shape(w, (1, 1), (1, 1)) and conv.conv_fwd_eligible(x, w, (1, 1), (1, 1))
    assert conv._fwd_lib().dtf_conv3x3_c64_ok(N, H, W)
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    y, part = conv.conv_fwd(x, w, (1, 1), (1, 1), stats=True)
    y0 = conv.conv_fwd(x, w, (1, 1), (1, 1))
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=2e-2)
    assert float((y.float() - ref).norm() / ref.norm()) < 5e-3
    assert torch.equal(y, y0)
    yf = y.float()
    torch.testing.assert_close(part[:, 0].sum(0), yf.sum((0, 2, 3)), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(part[:, 1].sum(0), (yf * yf).sum((0, 2, 3)), rtol=1e-4, atol=2e-2)


def test_conv3x3_c64_module_forward_backward(dev):
    """Conv2dTN(64, 64, 3) routed to K16 (forward, data gradient as a forward convolution of dY with the flipped filter,
    BatchNorm fed from the epilogue statistics) against the fp32 composition."""
    from distributed_tensorflow_amd.ops import batchnorm as bnmod, conv

    torch.manual_seed(0)
    x = torch.randn(4, 64, 20, 24, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    m = conv.Conv2dTN(64, 64, 3, stride=1, padding=1, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    bn = bnmod.FusedBatchNormAct(64, relu=True).to(dev).to(torch.bfloat16).train()
    x2, w2 = x.detach().float().requires_grad_(), m.weight.detach().float().requires_grad_()
    c2 = F.conv2d(x2, w2, None, 1, 1)
    r2 = F.relu(F.batch_norm(c2, None, None, bn.weight.detach().float(), bn.bias.detach().float(), True, 0.1, bn.eps))
    c, st = m.forward_stats(x)
    assert st is not None
    r = bn(c, stats=st)
    torch.testing.assert_close(r.float(), r2, rtol=3e-2, atol=3e-2)
    g = torch.randn_like(r2)
    r.backward(g.to(torch.bfloat16))
    r2.backward(g.to(torch.bfloat16).float())

    def rel(a, b):
        return float((a.float() - b).norm() / b.norm())

    assert rel(x.grad, x2.grad) < 3e-2
    assert rel(m.weight.grad, w2.grad) < 3e-2


def test_stem_pool_gather_with_bn_reduction_matches_separate_pass(dev, monkeypatch):
    """The K9 pool gather that also reduces for the BatchNorm backward vs pool gather + K7 reduction pass: same values
    summed in another order (gamma / beta / filter gradients to fp32 round-off of the partial sums)."""
    from distributed_tensorflow_amd.ops import batchnorm as bnmod, stem

    if not hasattr(stem._lib(), "dtf_stem_wrw"):
        pytest.skip("fused weight gradient not built")
    x, w = _inputs(dev, 5, 70, 96, seed=3)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    bn = bnmod.FusedBatchNormAct(64, relu=True).to(dev).to(torch.bfloat16).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    out = []
    for fused in (True, False):
        monkeypatch.setattr(stem, "_POOL_RED", fused)
        for p in (conv.weight, bn.weight, bn.bias):
            p.grad = None
        y = stem.stem_conv_bn_relu_pool(x, conv, bn, 3, 2, 1)
        torch.manual_seed(5)
        y.backward(torch.randn_like(y))
        out.append([p.grad.float().clone() for p in (conv.weight, bn.weight, bn.bias)])
    for a, b in zip(*out):
        assert float((a - b).norm() / b.norm()) < 1e-2


def test_shipped_gemm_solutions_are_numerically_right(dev):
    """Every recorded library GEMM solution (gemm_db/) against fp32 on NaN-poisoned memory (scripts/validate_gemms.py, own
    process: TunableOp is configured before torch is imported).  One recorded solution (the 64 -> 256 channel 1x1 forward
    product at batch 256) returned non-finite values, which BatchNorm + ReLU then silently turned into zeros."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "validate_gemms.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_bench_network_trains_at_the_bench_batch_size(dev):
    """bench.py's own loop (ResNet-50, batch 256, ShardedPS, the shipped MIOpen / GEMM solution files) for ten steps: every
    gradient and parameter finite, and the loss on the fixed batch falls.  The whole-model tests run at batch sizes whose
    library kernels are picked by heuristics; only this one sees the tuned batch-256 solutions."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "nan_probe_ps.py"), "256", "10"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    steps = re.findall(r"step (\d+) loss ([0-9.naninf]+) non-finite gradients (\d+) .* parameters (\d+)", r.stdout)
    assert len(steps) == 10, r.stdout[-2000:]
    assert all(int(g) == 0 and int(p) == 0 for _, _, g, p in steps), r.stdout[-2000:]
    losses = [float(x) for _, x, _, _ in steps]
    assert min(losses[5:]) < losses[0] - 0.3, losses


def test_resnet50_layer_shapes_at_bench_batch_match_fp32(dev):
    """Every convolution layer shape of ResNet-50 at batch 256 through the model's own routing (own kernels, tuned MIOpen
    solvers, recorded GEMM solutions): forward, data gradient, weight gradient vs fp32 on NaN-poisoned memory
    (scripts/validate_convs.py, own process)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "validate_convs.py"), "256"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_inception_v3_layer_shapes_at_bench_batch_match_fp32(dev):
    """the same validation for every convolution module geometry Inception-v3 meets at batch 256
    (recorded from one forward pass of the model: 1x7 / 7x1 / 3x3 / 5x5 / fused-head 1x1 layers on K13, K14, K6T, tuned
    MIOpen and the recorded GEMM solutions, weight gradients on the side stream): forward, data gradient, weight
    gradient vs fp32 on NaN-poisoned memory."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "validate_convs.py"), "256", "inception_v3"],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-1000:]
    assert "0 of" in r.stdout.splitlines()[-1]


# ---------------------------------------------------------------------------------------------------------------------
# K17: streaming 1x1 convolution (64 -> 256 channels) with the statistics epilogue (ops/csrc/conv1x1_stats_kernels.hip)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [(32, 56, 56), (5, 115, 114), (3, 149, 147)])
def test_conv1x1_k17_matches_reference(dev, cfg):
    """Forward + statistics partials vs fp32 (pixel counts that are / are not a multiple of the 128-pixel tile), then the
    module path: Conv1x1.forward_stats -> FusedBatchNormAct(stats=...) forward and backward vs the fp32 composition."""
    from distributed_tensorflow_amd.ops import batchnorm as bnmod, conv

    N, H, W = cfg
    torch.manual_seed(0)
    x = torch.randn(N, 64, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    m = conv.Conv1x1(64, 256).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    bn = bnmod.FusedBatchNormAct(256, relu=True).to(dev).to(torch.bfloat16).train()
    assert conv._k17_ok(x, m.weight)
    y, part = m.forward_stats(x)
    assert part is not None and "K17" in type(y.grad_fn).__name__
    x2, w2 = x.detach().float().requires_grad_(), m.weight.detach().float().requires_grad_()
    c2 = F.conv2d(x2, w2)
    assert y.shape == c2.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.detach().float() - c2.detach()).norm() / c2.detach().norm()) < 5e-3
    yf = y.detach().float()
    torch.testing.assert_close(part[:, 0].sum(0), yf.sum((0, 2, 3)), rtol=1e-4, atol=5e-2)
    torch.testing.assert_close(part[:, 1].sum(0), (yf * yf).sum((0, 2, 3)), rtol=1e-4, atol=5e-2)
    r = bn(y, stats=part)
    r2 = F.relu(F.batch_norm(c2, None, None, bn.weight.detach().float(), bn.bias.detach().float(), True, 0.1, bn.eps))
    torch.testing.assert_close(r.float(), r2, rtol=3e-2, atol=3e-2)
    g = torch.randn_like(r2)
    r.backward(g.to(torch.bfloat16))
    r2.backward(g.to(torch.bfloat16).float())
    assert float((x.grad.float() - x2.grad).norm() / x2.grad.norm()) < 3e-2
    assert float((m.weight.grad.float() - w2.grad).norm() / w2.grad.norm()) < 3e-2


def test_bn2_node_with_k17_partials(dev):
    """relu(BN_a(conv_a(x)) + BN_b(conv_b(x))) with both 64 -> 256 products on K17 and their statistics partials handed to
    the dual-BatchNorm node (dtf_bn2_fwd_train_pre) == the same node computing its own statistics; running statistics too."""
    from distributed_tensorflow_amd.ops import batchnorm as bnmod, conv

    torch.manual_seed(1)
    N, H, W = 32, 56, 56
    x = torch.randn(N, 64, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ca = conv.Conv1x1(64, 256).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    cb = conv.Conv1x1(64, 256).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    out = []
    for pre in (True, False, "a"):
        bna = bnmod.FusedBatchNormAct(256, relu=True).to(dev).to(torch.bfloat16).train()
        bnb = bnmod.FusedBatchNormAct(256, relu=False).to(dev).to(torch.bfloat16).train()
        with torch.no_grad():
            bna.weight.copy_(torch.linspace(0.5, 1.5, 256)); bnb.bias.copy_(torch.linspace(-0.3, 0.3, 256))
        xi = x.clone().requires_grad_()
        ca.weight.grad = cb.weight.grad = None
        ya, pa = ca.forward_stats(xi)
        yb, pb = cb.forward_stats(xi)
        assert pa is not None and pb is not None
        if pre is True:
            r = bnmod.fused_bn2_add_relu(ya, bna, yb, bnb, pa, pb)
        elif pre == "a":
            r = bnmod.fused_bn2_add_relu(ya, bna, yb, bnb, pa, None)
        else:
            r = bnmod.fused_bn2_add_relu(ya, bna, yb, bnb)
        assert "BN2" in type(r.grad_fn).__name__
        r.backward(torch.ones_like(r) * 0.01)
        out.append((r.detach().float(), xi.grad.float(), bna.running_var.float().clone(), bnb.running_mean.float().clone()))
    for o in (out[0], out[2]):
        torch.testing.assert_close(o[0], out[1][0], rtol=2e-2, atol=2e-2)
        assert float((o[1] - out[1][1]).norm() / out[1][1].norm()) < 1e-2
        torch.testing.assert_close(o[2], out[1][2], rtol=1e-2, atol=1e-3)
        torch.testing.assert_close(o[3], out[1][3], rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("k", [64, 256])
@pytest.mark.parametrize("rows", [65536 + 16, 16 * 12345])
def test_conv1x1_n64_k18_matches_reference(dev, k, rows):
    """K18 (register-streaming product onto 64 channels) vs fp32 on NaN-poisoned output memory; then a Conv1x1 k -> 64
    forward and a 64 -> k data gradient through the module path vs the fp32 composition."""
    from distributed_tensorflow_amd.ops import conv

    torch.manual_seed(k + rows)
    x2 = torch.randn(rows, k, device=dev).to(torch.bfloat16)
    w2 = (torch.randn(64, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    assert conv.mm_n64_ok(x2, k, 64)
    y = conv.mm_n64(x2, w2)
    ref = x2.float() @ w2.float().t()
    assert torch.isfinite(y).all()
    assert float((y.float() - ref).norm() / ref.norm()) < 5e-3
    torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=2e-2)


def test_conv1x1_n64_module_paths(dev):
    from distributed_tensorflow_amd.ops import conv

    torch.manual_seed(3)
    N, H, W = 32, 56, 56
    for cin, cout in ((256, 64), (64, 256), (64, 64)):
        x = torch.randn(N, cin, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
        m = conv.Conv1x1(cin, cout).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
        y = m(x)
        x2, w2 = x.detach().float().requires_grad_(), m.weight.detach().float().requires_grad_()
        c2 = F.conv2d(x2, w2)
        assert float((y.detach().float() - c2.detach()).norm() / c2.detach().norm()) < 5e-3
        g = torch.randn_like(c2).to(torch.bfloat16)
        y.backward(g)
        c2.backward(g.float())
        assert float((x.grad.float() - x2.grad).norm() / x2.grad.norm()) < 5e-3
        assert float((m.weight.grad.float() - w2.grad).norm() / w2.grad.norm()) < 2e-2


@pytest.mark.parametrize("cout,cin", [(256, 64), (64, 256)])
@pytest.mark.parametrize("N,H", [(64, 56), (24, 56), (8, 128)])
def test_k19_streaming_1x1_weight_gradient_matches_fp32(dev, cout, cin, N, H):
    """K19 (``conv1x1_wrw_kernels.hip``): dW of the 64 <-> 256-channel 1x1 layers as a pixel-streaming kernel
    (LDS-DMA tiles, transposing LDS reads, per-block fp32 slabs added in a fixed order).  vs the fp32 product of the
    same bf16 operands; bitwise repeatable; pixel counts that leave some of the 256 blocks without a tile."""
    from distributed_tensorflow_amd.ops import conv

    torch.manual_seed(N + cout)
    x = torch.randn(N, cin, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = (torch.randn(N, cout, H, H, device=dev, dtype=torch.bfloat16) * 0.1).contiguous(memory_format=torch.channels_last)
    dy2, x2 = conv._rows(dy), conv._rows(x)
    if not conv.wrw_stream_ok(dy2, x2):
        pytest.skip("shape not eligible for K19")
    got = conv.wrw_stream(dy2, x2)
    assert got.shape == (cout, cin) and got.dtype == torch.bfloat16
    ref = torch.zeros(cout, cin, device=dev, dtype=torch.float32)
    for lo in range(0, dy2.shape[0], 1 << 16):
        ref += dy2[lo:lo + (1 << 16)].float().t() @ x2[lo:lo + (1 << 16)].float()
    err = (got.float() - ref).abs().max().item()
    assert err <= 1e-2 * ref.abs().max().item() + 1e-3, err
    assert torch.equal(got, conv.wrw_stream(dy2, x2))
    # through the module path: the 1x1 convolution's autograd node routes its weight gradient here
    w = (torch.randn(cout, cin, 1, 1, device=dev, dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last)
    dw = conv._wrw(dy, x, w)
    assert torch.equal(dw.reshape(cout, cin), got)
