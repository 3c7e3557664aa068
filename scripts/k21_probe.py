"""K21 first-call probe: image_wrw on NaN-poisoned memory, first call vs second call vs fp32 (Inception-v3 stem shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from distributed_tensorflow_amd.ops import conv, hip
hip.lib()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(NB, 3, 299, 299, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = torch.randn(32, 3, 3, 3, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
dy = torch.randn(NB, 32, 149, 149, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
poison = torch.full((1 << 30,), float("nan"), dtype=torch.bfloat16, device=dev)      # 2 GB of NaN, then back to the pool
del poison
outs = []
for i in range(3):
    g = conv.image_wrw(dy, x, w, (2, 2), (0, 0))
    torch.cuda.synchronize()
    outs.append(g.clone())
    print("call", i, "finite", bool(torch.isfinite(g.float()).all()), "norm %.4f" % float(g.float().norm()), flush=True)
print("0 == 1:", torch.equal(outs[0], outs[1]), " 1 == 2:", torch.equal(outs[1], outs[2]))
_, ref, _ = torch.ops.aten.convolution_backward(dy[:8].float(), x[:8].float(), w.float(), None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
g8 = conv.image_wrw(dy[:8].contiguous(memory_format=torch.channels_last), x[:8].contiguous(memory_format=torch.channels_last), w, (2, 2), (0, 0))
print("vs fp32 (8 images): %.2e" % float((g8.float() - ref).norm() / ref.norm()))
