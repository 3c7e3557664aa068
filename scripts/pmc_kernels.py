"""Tiny driver that launches each hand-written kernel family a few times on representative
shapes (random data) so that rocprofv3 --pmc can attribute counters per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import batchnorm as bn, embedding as emb, linear as lin, optim

dev = torch.device("cuda")
torch.manual_seed(0)
# K6: 4096^3 and a BERT FFN shape
for (M, N, K) in [(4096, 4096, 4096), (8192, 3072, 768)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    for _ in range(3):
        lin.gemm_bias_act(x, w, b, "gelu")
# K7: ResNet layer-3 block output
shp = (256, 1024, 14, 14)
x = torch.randn(shp, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
r = torch.randn(shp, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
g, be = torch.ones(1024, device=dev, dtype=torch.bfloat16, requires_grad=True), torch.zeros(1024, device=dev, dtype=torch.bfloat16, requires_grad=True)
rm, rv = torch.zeros(1024, device=dev), torch.ones(1024, device=dev)
for _ in range(3):
    y = bn.fused_bn_act(x, g, be, r, rm, rv, True, 0.1, 1e-5, True)
    y.backward(torch.randn_like(y))
# K1 / K2 on a 25.6M-element shard (ResNet-50 at W=1)
n = 25_600_000
gr = torch.randn(n, device=dev).to(torch.bfloat16)
p, m, v, o = torch.randn(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.empty(n, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    optim.sgd_momentum_apply(gr, p, m, o, lr=0.1, momentum=0.9, wd=1e-4)
    optim.adam_apply(gr, p, m, v, o, lr=1e-3, step=3)
# K3 / K4: 2.6M-row fp32 table (1.3 GB, far beyond the 256 MiB Infinity Cache), 1.7M lookups
R, D, nidx = 5_000_000, 64, 65536 * 26
table = torch.randn(R, D, device=dev)
acc = torch.full((R, D), 0.1, device=dev)
idx = (R * torch.rand(nidx, device=dev).pow(3)).long().clamp_(0, R - 1)
for _ in range(3):
    out = emb.embedding_gather(table, idx, torch.bfloat16)
    emb.embedding_sparse_apply(table, acc, idx, out, opt="adagrad", lr=0.01)
torch.cuda.synchronize()
print("done")
