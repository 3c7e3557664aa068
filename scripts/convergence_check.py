"""Does the whole stack LEARN something that generalises?  bench.py's fixed batch can be memorised by almost any update
rule; here ResNet-50 (10 classes) trains on a generated dataset -- class k = a low-frequency pattern of its own under heavy
pixel noise, different every image -- fed through the input pipeline (memory-mapped uint8 file, sharded stream,
prefetcher), through ShardedPS with a live 1-rank communicator, weight-gradient side stream and pull prefetch, and is then
evaluated (BatchNorm in inference mode: the moving statistics the training kernels maintained) on images it never saw.
usage: convergence_check.py [steps] [batch] [resnet50 | inception_v3 | wide_deep | bert]

``bert``: a 4-layer BERT (hidden 256, 4 heads of 64: the K11 attention kernel, K10 LayerNorm, K12 cross-entropy, K6T, the
clipped sharded AdamW path) on generated sentences that alternate between two words (a random word of a 2 000-word
vocabulary and its successor) with 15 % of the positions masked: a masked word is the word two positions away, so the
model has to learn to attend by relative position and copy; the next-sentence label is the parity of the first word.
Masked-word accuracy on sentences it never saw (a 4-layer post-LN BERT did not learn the harder "counting" variant of
this -- every word the predecessor's successor -- within 3 000 steps at any rate tried, while
test_bert_fused_graph_matches_the_plain_fp32_model shows its gradients are the plain model's).

``wide_deep``: config #3 on a stream of fresh synthetic batches (Zipf-skewed ids; the label is the XOR of the parity of
feature 0's id and the sign of dense feature 0, so it can only be predicted through a trained embedding row AND the deep
tower), routed embeddings + sparse Adagrad on the tables, Adam on the tower; accuracy on a batch it never saw."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
import numpy as np
import torch
from distributed_tensorflow_amd.models import get_model_module
from distributed_tensorflow_amd.parallel.comm import Comm
from distributed_tensorflow_amd.parallel.ps_runtime import OptimizerSpec, ShardedPS
from distributed_tensorflow_amd.train import data as D

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
which = sys.argv[3] if len(sys.argv) > 3 else "resnet50"
if which == "bert":
    comm = Comm.from_env(force=True)
    dev = comm.device
    mod = get_model_module("bert")
    torch.manual_seed(0)
    V, S = 3000, 128
    m = mod.prepare(mod.build(layers=4, hidden=256, heads=4, intermediate=1024, vocab=V), dev).train()
    opt = mod.default_optimizer()
    peak = float(os.environ.get("DTF_CONV_LR", "3e-4"))
    opt.lr = peak
    ps = ShardedPS(list(m.named_parameters()), comm, opt)
    ps.attach(m)

    def sentences(n, seed):
        g = torch.Generator().manual_seed(seed)
        start = torch.randint(0, 2000, (n, 1), generator=g)
        # a sentence alternates between two words, start and start + 1: a masked word is the word two positions away
        ids = 1000 + (start + (torch.arange(S).unsqueeze(0) % 2)) % 2000
        types = (torch.arange(S).unsqueeze(0) >= S // 2).long().expand(n, -1).contiguous()
        npred = 19
        pos = torch.stack([torch.randperm(S, generator=g)[:npred] for _ in range(n)])
        labels = torch.gather(ids, 1, pos)
        ids = ids.scatter(1, pos, 103)
        return tuple(t.to(dev) for t in (ids, types, torch.ones(n, S, dtype=torch.long), pos, labels, (start.squeeze(1) % 2)))

    held = sentences(256, 10 ** 6)

    def acc_bert():
        m.eval()
        with torch.no_grad():
            mlm, nsp = m(held[0], held[1], None, held[3])
        m.train()
        return (float((mlm.float().argmax(-1) == held[4].reshape(-1)).float().mean()),
                float((nsp.float().argmax(-1) == held[5]).float().mean()))

    a0 = acc_bert()
    ls = []
    for it in range(steps):
        ps.set_lr(peak * min(1.0, (it + 1) / 100.0))
        loss = mod.loss_fn(m, sentences(B, it))
        loss.backward()
        ps.step()
        ls.append(loss.detach())
        if (it + 1) % 500 == 0:
            print("   step %d loss %.3f accuracy %s" % (it + 1, float(loss), acc_bert()), flush=True)
    torch.cuda.synchronize()
    ls = [float(v) for v in ls]
    a1 = acc_bert()
    print("bert (4 layers) steps %d batch %d seq %d: loss %.3f -> %.3f; held-out masked-word accuracy %.3f -> %.3f, next-sentence %.3f -> %.3f" % (
        steps, B, S, ls[0], sum(ls[-10:]) / 10, a0[0], a1[0], a0[1], a1[1]), flush=True)
    comm.close()
    sys.exit(0 if a1[0] > 0.8 else 1)
if which == "wide_deep":
    comm = Comm.from_env(force=True)
    dev = comm.device
    mod = get_model_module("wide_deep")
    torch.manual_seed(0)
    m = mod.prepare(mod.build(comm=comm, device=dev), dev).train()
    ps = ShardedPS(list(m.named_parameters()), comm, mod.default_optimizer())
    ps.attach(m)
    held = mod.synthetic_batch(B, device=dev, seed=10 ** 6, model=m)

    def acc_wd():
        m.eval()
        with torch.no_grad():
            p = m(held[0], held[1]).squeeze(1).float() > 0
        m.train()
        return float((p == (held[2] > 0.5)).float().mean())

    a0 = acc_wd()
    ls = []
    for it in range(steps):
        loss = mod.loss_fn(m, mod.synthetic_batch(B, device=dev, seed=it, model=m))
        loss.backward()
        ps.step()
        ls.append(loss.detach())
    torch.cuda.synchronize()
    ls = [float(v) for v in ls]
    a1 = acc_wd()
    print("wide_deep steps %d batch %d: loss %.4f -> %.4f; held-out accuracy %.3f -> %.3f (chance 0.500)" % (
        steps, B, ls[0], sum(ls[-10:]) / 10, a0, a1), flush=True)
    comm.close()
    sys.exit(0 if a1 > 0.8 else 1)
S, NTRAIN, NTEST, K = (96 if which == "inception_v3" else 64), (3072 if which == "inception_v3" else 6144), 1024, 10


def make(n, seed):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, K, n)
    yy, xx = np.meshgrid(np.linspace(0, 1, S), np.linspace(0, 1, S), indexing="ij")
    base = np.stack([np.sin(2 * np.pi * ((k % 5 + 1) * xx + (k // 5 + 1) * yy))[..., None] * np.array([1.0, -0.5, 0.25])[(np.arange(3) + k) % 3]
                     for k in range(K)])                                   # [K, S, S, 3]
    img = 128 + 40 * base[y] + 45 * rng.standard_normal((n, S, S, 3))       # signal well below the noise per pixel
    return np.clip(img, 0, 255).astype(np.uint8), y.astype(np.int64)


d = tempfile.mkdtemp(prefix="dtf-conv-")
xtr, ytr = make(NTRAIN, 0)
xte, yte = make(NTEST, 1)
np.save(os.path.join(d, "images.npy"), xtr)
np.save(os.path.join(d, "labels.npy"), ytr)
comm = Comm.from_env(force=True)
dev = comm.device
mod = get_model_module(which)
torch.manual_seed(0)
m = mod.prepare(mod.build(num_classes=K), dev).train()
mode, stale_k = os.environ.get("DTF_UPDATE_MODE", "sync"), int(os.environ.get("DTF_STALENESS", "1"))
spec = OptimizerSpec(kind="momentum", lr=0.05, momentum=0.9, weight_decay=1e-4)
if mode == "async":
    from distributed_tensorflow_amd.parallel.async_ps import AsyncPS

    ps = AsyncPS(list(m.named_parameters()), comm, spec)
else:
    ps = ShardedPS(list(m.named_parameters()), comm, spec, update_mode=mode, staleness=stale_k)
    if mode == "sync":
        ps.attach(m)
ds = D.open_dataset(d)
pf = D.Prefetcher(D.BatchStream(ds, B, seed=0), D.transform_for(mod, ds, dev), dev)
tf = D.transform_for(mod, ds, dev)


def accuracy(x, y):
    m.eval()
    hit = 0
    with torch.no_grad():
        for i in range(0, len(x), 256):
            bx, by = tf([torch.from_numpy(x[i:i + 256]).to(dev), torch.from_numpy(y[i:i + 256]).to(dev)])
            hit += int((m(bx).float().argmax(1) == by).sum())
    m.train()
    return hit / len(x)


acc0 = accuracy(xte, yte)
losses = []
for it in range(steps):
    ps.set_lr(0.05 * min(1.0, (it + 1) / 30.0))
    loss = mod.loss_fn(m, next(pf))
    loss.backward()
    ps.step()
    losses.append(loss.detach())
pf.close()
torch.cuda.synchronize()
ls = [float(v) for v in losses]
acc = accuracy(xte, yte)
print("%s %s%s steps %d batch %d: loss %.3f -> %.3f (mean of last 10: %.3f); held-out accuracy %.3f -> %.3f (chance 0.100)" % (
    which, mode, ("-%d" % stale_k) if mode == "stale" else "", steps, B, ls[0], ls[-1], sum(ls[-10:]) / 10, acc0, acc), flush=True)
comm.close()
sys.exit(0 if acc > 0.8 else 1)
