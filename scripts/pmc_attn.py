import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import attention as att
qkv = torch.randn(64, 512, 3, 12, 64, device="cuda").to(torch.bfloat16)
do = torch.randn(64, 512, 768, device="cuda").to(torch.bfloat16)
for _ in range(3):
    o, lse = att.attn_fwd_packed(qkv)
    att.attn_bwd_packed(qkv, o, lse, do)
torch.cuda.synchronize()
print("done")
