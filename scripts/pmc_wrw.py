"""K6T-conv (weight gradient) on the narrow ResNet layers for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import conv, hip
hip.lib()
dev = torch.device("cuda")
torch.manual_seed(0)
for (NB, C, H, Co) in [(256, 64, 56, 64), (256, 128, 28, 128), (256, 256, 14, 256)]:
    x = torch.randn(NB, C, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, C, 3, 3, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(NB, Co, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        conv.conv_wrw(dy, x, w, (1, 1), (1, 1))
        torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
torch.cuda.synchronize()
print("done")
