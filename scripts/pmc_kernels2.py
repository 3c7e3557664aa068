"""Second PMC driver: the kernels added after profiles/r1_03 (K6 persistent tile, K6T plain / conv, dual BatchNorm,
BN inside the pool, K9, K10, K11, K12, embedding plan), each a few times on a representative shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen
miopen.setup()
import torch
from distributed_tensorflow_amd.ops import attention, batchnorm as bn, conv, cross_entropy as ce, embedding as emb, layernorm, linear as lin, pool

dev = torch.device("cuda")
torch.manual_seed(0)
cl = torch.channels_last


def rnd(*s):
    return torch.randn(*s, device=dev).to(torch.bfloat16)


# K6 persistent 256 tile (BERT FFN1 with GELU + saved pre-activation) and K6T with fused bias gradient
x, w, b = rnd(32768, 768), rnd(3072, 768) * 0.03, rnd(3072)
dz = rnd(32768, 3072)
for _ in range(3):
    lin.gemm_bias_act(x, w, b, "gelu_tanh", None, save_preact=True)
    lin.gemm_tn(dz, x, colsum_dtype=torch.bfloat16)
# K6T conv variant: ResNet 256ch 3x3 at 14x14, batch 256
xi = rnd(256, 256, 14, 14).contiguous(memory_format=cl)
wi = rnd(256, 256, 3, 3).contiguous(memory_format=cl)
dyi = rnd(256, 256, 14, 14).contiguous(memory_format=cl)
for _ in range(3):
    conv.conv_wrw(dyi, xi, wi, (1, 1), (1, 1))
# dual BatchNorm node (ResNet stage-2 first block) and BN inside the stem pool
shp = (256, 512, 28, 28)
bna = bn.FusedBatchNormAct(512, relu=True).to(dev).to(torch.bfloat16).train()
bnb = bn.FusedBatchNormAct(512, relu=False).to(dev).to(torch.bfloat16).train()
xa, xb = rnd(*shp).contiguous(memory_format=cl).requires_grad_(), rnd(*shp).contiguous(memory_format=cl).requires_grad_()
for _ in range(3):
    y = bn.fused_bn2_add_relu(xa, bna, xb, bnb)
    y.backward(torch.randn_like(y))
bn1 = bn.FusedBatchNormAct(64, relu=True).to(dev).to(torch.bfloat16).train()
xs = rnd(256, 64, 112, 112).contiguous(memory_format=cl).requires_grad_()
for _ in range(3):
    y = pool.bn_relu_max_pool2d(xs, bn1, 3, 2, 1)
    y.backward(torch.randn_like(y))
# K9 average pool (Inception 35x35x64 head slice), K10 LayerNorm, K11 attention, K12 cross-entropy
xp = rnd(256, 64, 35, 35).contiguous(memory_format=cl).requires_grad_()
for _ in range(3):
    y = pool.avg_pool2d(xp, 3, 1, 1, count_include_pad=False)
    y.backward(torch.randn_like(y))
ln = layernorm.FusedLayerNorm(768, eps=1e-12).to(dev).to(torch.bfloat16)
xl = rnd(32768, 768).requires_grad_()
qkv = rnd(64, 512, 3, 12, 64).requires_grad_()
logits = rnd(4864, 30522).requires_grad_()
labels = torch.randint(0, 30522, (4864,), device=dev)
for _ in range(3):
    y = ln(xl)
    y.backward(torch.randn_like(y))
    o = attention.attention_packed(qkv)
    o.backward(torch.randn_like(o))
    ce.cross_entropy(logits, labels).backward()
# embedding segment plan + dense embedding gradient (BERT word embeddings)
idx = (30522 * torch.rand(64, 512, device=dev).pow(2)).long().clamp_(0, 30521)
we = rnd(30522, 768).requires_grad_()
for _ in range(3):
    emb.dense_embedding(idx, we).backward(rnd(64, 512, 768))
torch.cuda.synchronize()
print("done")
