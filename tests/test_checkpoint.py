"""Checkpoint layout contract: shard bytes are storage order, the index records it, ``read`` returns
LOGICAL tensors and ``restore_ps`` converts between memory formats.  Plus BERT's packed QKV projection listed as
google-research/bert's query / key / value variables."""
import torch
import torch.nn as nn

from distributed_tensorflow_amd.parallel import checkpoint as ckpt
from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS


class _ConvNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 5, 3, bias=False)
        self.conv2 = nn.Conv2d(5, 4, (1, 7), bias=False)
        self.fc = nn.Linear(4, 2)


def _train_a_bit(model, ps):
    g = torch.Generator().manual_seed(3)
    for _ in range(2):
        for p in ps.params:
            p.grad = torch.randn(p.shape, generator=g).contiguous(memory_format=torch.channels_last) \
                if p.dim() == 4 and not p.is_contiguous() else torch.randn(p.shape, generator=g)
        ps.step()


def test_channels_last_checkpoint_reads_back_logical_and_restores_into_contiguous(tmp_path):
    torch.manual_seed(0)
    a = _ConvNet().to(memory_format=torch.channels_last)
    assert not a.conv.weight.is_contiguous()
    opt = OptimizerSpec(kind="momentum", lr=0.1)
    ps_a = ShardedPS(list(a.named_parameters()), opt=opt, wire_dtype=torch.float32)
    _train_a_bit(a, ps_a)
    ckpt.save_ps(ps_a, str(tmp_path), tf_names={"conv.weight": "net/conv/weights"})
    rd = ckpt.CheckpointReader(str(tmp_path))
    assert rd.index["variables"]["net/conv/weights"]["storage_order"] == [0, 2, 3, 1]
    assert "storage_order" not in rd.index["variables"]["fc.weight"]
    # read() returns the logical [O, I, H, W] tensor, whatever the byte order in the shard
    torch.testing.assert_close(rd.read("net/conv/weights"), a.conv.weight.detach(), rtol=0, atol=0)
    torch.testing.assert_close(rd.read("conv2.weight"), a.conv2.weight.detach(), rtol=0, atol=0)
    # restore into a CONTIGUOUS model (a GPU/NHWC checkpoint into a CPU job): filters must not be scrambled
    torch.manual_seed(1)
    b = _ConvNet()
    ps_b = ShardedPS(list(b.named_parameters()), opt=opt, wire_dtype=torch.float32)
    ckpt.restore_ps(ps_b, str(tmp_path), tf_names={"conv.weight": "net/conv/weights"})
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        torch.testing.assert_close(pb.detach(), pa.detach(), rtol=0, atol=0, msg=n)
    # momentum slots too, and the reverse direction (contiguous checkpoint -> channels_last model)
    torch.testing.assert_close(rd.read("net/conv/weights/Momentum").contiguous().reshape(-1),
                               ps_b.param_from_flat(ps_b.full_master() * 0 + _full_slot(ps_b), 0).reshape(-1),
                               rtol=0, atol=0)
    ckpt.save_ps(ps_b, str(tmp_path / "b"))
    torch.manual_seed(2)
    c = _ConvNet().to(memory_format=torch.channels_last)
    ps_c = ShardedPS(list(c.named_parameters()), opt=opt, wire_dtype=torch.float32)
    ckpt.restore_ps(ps_c, str(tmp_path / "b"))
    for (n, pa), (_, pc) in zip(a.named_parameters(), c.named_parameters()):
        torch.testing.assert_close(pc.detach(), pa.detach(), rtol=0, atol=0, msg=n)
        assert pc.stride() == pa.stride()


def _full_slot(ps):
    out = torch.zeros(ps.layout.total)
    for bk in ps.layout.buckets:
        out[bk.start:bk.start + bk.size] = ps.slot_m[bk.shard_off:bk.shard_off + bk.shard]
    return out


def test_bert_qkv_is_listed_as_query_key_value(tmp_path):
    from distributed_tensorflow_amd.models import bert

    torch.manual_seed(0)
    m = bert.build(layers=2, hidden=32, heads=2, intermediate=64, vocab=100)
    names = bert.tf_variable_names(m)
    ps = ShardedPS(list(m.named_parameters()), opt=bert.default_optimizer(), wire_dtype=torch.float32, bucket_mb=0.004)
    assert len(ps.layout.buckets) > 3
    ckpt.save_ps(ps, str(tmp_path), tf_names=names)
    rd = ckpt.CheckpointReader(str(tmp_path))
    w = m.layers[1].qkv.weight.detach()
    for i, part in enumerate(("query", "key", "value")):
        k = rd.read("bert/encoder/layer_1/attention/self/%s/kernel" % part)
        assert tuple(k.shape) == (32, 32)
        torch.testing.assert_close(k, w[32 * i:32 * (i + 1)], rtol=0, atol=0)
        b = rd.read("bert/encoder/layer_1/attention/self/%s/bias" % part)
        torch.testing.assert_close(b, m.layers[1].qkv.bias.detach()[32 * i:32 * (i + 1)], rtol=0, atol=0)
        assert rd.has("bert/encoder/layer_1/attention/self/%s/kernel/Adam_1" % part)
    assert not any("qkv" in n for n in rd.names())
    # and back: a differently initialised model restores bit-exactly through the split names
    torch.manual_seed(9)
    m2 = bert.build(layers=2, hidden=32, heads=2, intermediate=64, vocab=100)
    ps2 = ShardedPS(list(m2.named_parameters()), opt=bert.default_optimizer(), wire_dtype=torch.float32, bucket_mb=0.05)
    ckpt.restore_ps(ps2, str(tmp_path), tf_names=bert.tf_variable_names(m2))
    for (n, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        torch.testing.assert_close(p2.detach(), p1.detach(), rtol=0, atol=0, msg=n)


def test_old_checkpoints_are_pruned_like_tf_max_to_keep(tmp_path, monkeypatch):
    """``keep_checkpoints`` (TF1 Saver's max_to_keep): after every published index only the newest K complete checkpoints
    remain -- every file of the older ones is gone, shards of a step that has no index yet are left alone, the
    ``checkpoint`` file lists what is there, and the newest one restores."""
    import os

    monkeypatch.setattr(ckpt, "KEEP", 3)
    torch.manual_seed(0)
    model = _ConvNet()
    ps = ShardedPS(list(model.named_parameters()), None, OptimizerSpec(kind="momentum", lr=0.1, momentum=0.9))
    export = str(tmp_path)
    open(os.path.join(export, "model.ckpt-999.data-00000-of-00001"), "wb").write(b"in flight")   # no index: not ours to delete
    for step in range(1, 8):
        _train_a_bit(model, ps)
        ckpt.save_ps(ps, export, step=step, replicated={"global_step": torch.tensor(step)})
        assert ckpt.checkpoint_steps(export) == list(range(max(1, step - 2), step + 1))
    files = sorted(os.listdir(export))
    assert not any(f.startswith("model.ckpt-%d." % s) for s in (1, 2, 3, 4) for f in files), files
    assert all(any(f.startswith("model.ckpt-%d." % s) for f in files) for s in (5, 6, 7))
    assert "model.ckpt-999.data-00000-of-00001" in files
    text = open(os.path.join(export, "checkpoint")).read()
    assert 'model_checkpoint_path: "model.ckpt-7"' in text
    assert [l for l in text.splitlines() if l.startswith("all_model")] == ['all_model_checkpoint_paths: "model.ckpt-%d"' % s for s in (5, 6, 7)]
    assert ckpt.latest_step(export) == 7
    model2 = _ConvNet()
    ps2 = ShardedPS(list(model2.named_parameters()), None, OptimizerSpec(kind="momentum", lr=0.1, momentum=0.9))
    assert ckpt.restore_ps(ps2, export).step == 7
    assert torch.equal(ps2.master, ps.master)
    monkeypatch.setattr(ckpt, "KEEP", 0)                      # 0: keep everything
    for step in (8, 9, 10, 11):
        ckpt.save_ps(ps, export, step=step)
    assert ckpt.checkpoint_steps(export) == [5, 6, 7, 8, 9, 10, 11]


def test_inspect_checkpoint_tool_lists_prints_and_exports(tmp_path, capsys):
    """``python -m distributed_tensorflow_amd.tools.inspect_checkpoint`` (TF1's inspect_checkpoint for the `export`
    directory of a job): listing with TF-style names and optimizer slots, one tensor, one-file export that another
    program can load without this package."""
    from safetensors.torch import load_file

    from distributed_tensorflow_amd.models import mnist_mlp
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.tools import inspect_checkpoint as tool
    from distributed_tensorflow_amd.train import session as S

    d = str(tmp_path / "export")
    m = mnist_mlp.build()
    with S.MonitoredTrainingSession(m, "momentum", comm=Comm.single(), checkpoint_dir=d,
                                    hooks=[S.StopAtStepHook(last_step=3)], tf_names=mnist_mlp.tf_variable_names(m)) as sess:
        while not sess.should_stop():
            sess.run(mnist_mlp.loss_fn(m, mnist_mlp.synthetic_batch(8, device="cpu", seed=1)))
    assert tool.main([d]) == 0
    out = capsys.readouterr().out
    assert "step 3, written by 1 rank(s), optimizer momentum" in out and "hid_w/Momentum" in out and "[784, 100]" in out
    assert "global_step" in out and "8 variables, 159020 elements" in out
    assert tool.main([d, "--tensor_name", "sm_b"]) == 0
    assert "tensor_name: sm_b  shape [10]  dtype float32" in capsys.readouterr().out
    assert tool.main([d, "--tensor_name", "nope"]) == 1 and tool.main([str(tmp_path / "nothing")]) == 1
    capsys.readouterr()
    f = str(tmp_path / "model.safetensors")
    assert tool.main([d, "--export", f]) == 0
    t = load_file(f)
    assert sorted(t) == ["global_step", "hid_b", "hid_w", "sm_b", "sm_w"] and int(t["global_step"]) == 3
    names = mnist_mlp.tf_variable_names(m)
    for n, p in m.named_parameters():
        assert torch.equal(t[names[n]], p.detach()), n
    assert tool.main([d, "--export", str(tmp_path / "all.pt"), "--with_slots"]) == 0
    allt = torch.load(str(tmp_path / "all.pt"), weights_only=True)
    assert "hid_w/Momentum" in allt and allt["hid_w/Momentum"].shape == (784, 100)
