"""K11 attention vs torch SDPA (CK flash kernels) on the BERT-base shape."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import attention as att

dev = torch.device("cuda")
for B, S, A in [(64, 512, 12), (16, 512, 12), (64, 128, 12)]:
    qkv = torch.randn(B, S, 3, A, 64, device=dev).to(torch.bfloat16)
    do = torch.randn(B, S, A * 64, device=dev).to(torch.bfloat16)
    xg = qkv.clone().requires_grad_()
    o_k, lse_k = att.attn_fwd_packed(qkv)
    o_r = att.attention_reference(xg)
    fns = {"k11_fwd": lambda: att.attn_fwd_packed(qkv), "sdpa_fwd": lambda: att.attention_reference(qkv),
           "k11_bwd": lambda: att.attn_bwd_packed(qkv, o_k, lse_k, do),
           "sdpa_bwd": lambda: torch.autograd.grad(o_r, xg, do, retain_graph=True)}
    row = {"B": B, "S": S, "A": A}
    for k, f in fns.items():
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        row[k + "_us"] = round(t * 1e6, 1)
        row[k + "_TF"] = round(4.0 * B * A * S * S * 64 / t / 1e12, 1)
    print(json.dumps(row), flush=True)
