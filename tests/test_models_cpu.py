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
