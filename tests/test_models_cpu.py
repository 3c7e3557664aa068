"""CPU checks of model-level autograd helpers (no kernels involved)."""
import torch


def test_bert_split_qkv_gradient_matches_plain_autograd():
    from distributed_tensorflow_amd.models.bert import _SplitQKV

    torch.manual_seed(0)
    B, S, A, D = 2, 5, 3, 4
    x = torch.randn(B, S, 3, A, D, requires_grad=True)
    x2 = x.detach().clone().requires_grad_()
    gs = [torch.randn(B, A, S, D) for _ in range(3)]
    q, k, v = _SplitQKV.apply(x)
    q2, k2, v2 = x2.permute(2, 0, 3, 1, 4).unbind(0)
    for a, b in ((q, q2), (k, k2), (v, v2)):
        assert torch.equal(a, b)
    torch.autograd.backward([q, k, v], gs)
    torch.autograd.backward([q2, k2, v2], gs)
    assert torch.equal(x.grad, x2.grad)


def test_bert_tiny_forward_backward_cpu():
    from distributed_tensorflow_amd.models import bert

    torch.manual_seed(0)
    m = bert.build(layers=2, hidden=32, heads=4, intermediate=64, vocab=2000, seq_len=16)
    batch = bert.synthetic_batch(2, device="cpu", seed=1, seq_len=16, vocab=2000)
    loss = bert.loss_fn(m, batch)
    loss.backward()
    assert torch.isfinite(loss)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_wide_deep_split_gradient_matches_autograd_slices():
    from distributed_tensorflow_amd.models.wide_deep import _SplitDeepWide

    torch.manual_seed(0)
    B, F_, D, W, nd, kpad = 5, 4, 6, 9, 3, 5
    dense = torch.randn(B, nd)
    e1 = torch.randn(B, F_, W, requires_grad=True)
    e2 = e1.detach().clone().requires_grad_()
    h1, wide1 = _SplitDeepWide.apply(e1, dense, D, kpad)
    h2 = torch.cat([e2[:, :, :D].reshape(B, -1), dense, torch.zeros(B, kpad)], 1)
    wide2 = e2[:, :, D].float().sum(1, keepdim=True)
    assert torch.equal(h1, h2) and torch.equal(wide1, wide2)
    gh, gw = torch.randn_like(h2), torch.randn_like(wide2)
    torch.autograd.backward([h1, wide1], [gh, gw])
    torch.autograd.backward([h2, wide2], [gh, gw])
    assert torch.equal(e1.grad, e2.grad)
    # one output unused
    e3 = e1.detach().clone().requires_grad_()
    h3, _ = _SplitDeepWide.apply(e3, dense, D, 0)
    g3 = torch.randn_like(h3)
    h3.backward(g3)
    assert torch.equal(e3.grad[:, :, :D].reshape(B, -1), g3[:, :F_ * D]) and float(e3.grad[:, :, D:].abs().max()) == 0.0


def test_inception_pool_branch_commutes():
    """avg-pool (count_include_pad=False) and a 1x1 convolution commute: the order used on the GPU
    (models/inception_v3.py::_pool_branch) computes the same function as TF-slim's."""
    import torch.nn.functional as F

    torch.manual_seed(0)
    x = torch.randn(2, 12, 9, 7, dtype=torch.float64)
    w = torch.randn(5, 12, 1, 1, dtype=torch.float64)
    a = F.conv2d(F.avg_pool2d(x, 3, 1, 1, count_include_pad=False), w)
    b = F.avg_pool2d(F.conv2d(x, w), 3, 1, 1, count_include_pad=False)
    torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)


def test_subsample_fork_gradient():
    """Strided 1x1 shortcut as subsample + dense 1x1: same values and input gradient as the strided convolution."""
    import torch.nn.functional as F

    from distributed_tensorflow_amd.ops.conv import subsample_fork

    torch.manual_seed(0)
    x1 = torch.randn(2, 6, 9, 8, requires_grad=True)
    w = torch.randn(5, 6, 1, 1)
    w3 = torch.randn(4, 6, 3, 3)
    xs, xp = subsample_fork(x1 * 1.0, 2)
    out1 = F.conv2d(xs, w).sum() + (F.conv2d(xp, w3, padding=1) ** 2).sum()
    out1.backward()
    x2 = x1.detach().clone().requires_grad_()
    out2 = F.conv2d(x2, w, stride=2).sum() + (F.conv2d(x2, w3, padding=1) ** 2).sum()
    out2.backward()
    torch.testing.assert_close(out1, out2)
    torch.testing.assert_close(x1.grad, x2.grad)


def test_weight_gradient_is_brought_into_the_parameter_layout():
    """ops/streams._in_param_layout (the copy autograd would otherwise make on the wrong stream, see its docstring): a
    gradient with the parameter's strides passes through untouched, a strided view comes back dense in the parameter's
    layout (channels-last included) with the same values, and the diagnostic counter moves."""
    import torch

    from distributed_tensorflow_amd.ops import streams

    w = torch.randn(8, 3, 3, 3).contiguous(memory_format=torch.channels_last)
    same = torch.randn(8, 3, 3, 3).contiguous(memory_format=torch.channels_last)
    assert streams._in_param_layout(same, w) is same
    before = streams.layout_copies
    wide = torch.randn(8, 32)
    view = wide[:, :27].reshape(8, 3, 3, 3).permute(0, 3, 1, 2)          # strides (32, 1, 9, 3): K21's first version
    assert view.stride() != w.stride()
    out = streams._in_param_layout(view, w)
    assert out.stride() == w.stride() and torch.equal(out, view) and out.data_ptr() != view.data_ptr()
    assert streams.layout_copies == before + 1
    assert streams._in_param_layout(None, w) is None
    other_shape = torch.randn(4, 27)
    assert streams._in_param_layout(other_shape, w) is other_shape
