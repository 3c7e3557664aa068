"""K13 driver for rocprofv3 --pmc: the implicit-GEMM convolution on ResNet-50 / Inception-v3 layer shapes (random data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import conv, hip

hip.lib()
dev = torch.device("cuda")
torch.manual_seed(0)
for (NB, C, H, Co, KH, KW, st, pad) in [(256, 256, 14, 256, 3, 3, 1, (1, 1)), (256, 512, 7, 512, 3, 3, 1, (1, 1)),
                                        (256, 192, 17, 192, 1, 7, 1, (0, 3)), (256, 448, 8, 384, 3, 3, 1, (1, 1))]:
    x = torch.randn(NB, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, C, KH, KW, device=dev) * (C * KH * KW) ** -0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        conv.conv_fwd(x, w, (st, st), pad, stats=True)
torch.cuda.synchronize()
print("done")
