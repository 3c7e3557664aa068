"""K1 / K2 shard-apply rates on shard sizes of the model families (bf16 gradient in, fp32 master + slots, bf16 pull copy
out).  Cold: three rotating shard sets larger than the Infinity Cache where the size allows.  Prints one JSON line per
size; DTF_STREAM_GRID_CAP selects the grid shape (2048 = grid-strided default, a huge value = one block per 2048
elements)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributed_tensorflow_amd.ops import optim

dev = torch.device("cuda")


def bench(fn, sets, iters=12):
    for i in range(3):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(sets[i % len(sets)])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for name, n in [("ResNet-50 W=1 (25.6 M)", 25_600_000), ("BERT-base W=8 shard (13.7 M)", 13_700_000),
                ("BERT-base W=1 (110 M)", 110_000_000), ("1 G elements", 1 << 30)]:
    nset = 1 if n >= (1 << 29) else 3
    sets = []
    for _ in range(nset):
        sets.append(dict(g=torch.randn(n, device=dev).to(torch.bfloat16), p=torch.randn(n, device=dev),
                         m=torch.zeros(n, device=dev), v=torch.zeros(n, device=dev),
                         o=torch.empty(n, device=dev, dtype=torch.bfloat16)))
    sgd = bench(lambda s: optim.sgd_momentum_apply(s["g"], s["p"], s["m"], s["o"], lr=0.1, momentum=0.9, wd=1e-4), sets)
    adam = bench(lambda s: optim.adam_apply(s["g"], s["p"], s["m"], s["v"], s["o"], lr=1e-4, wd=0.01, step=3), sets)
    print(json.dumps({"shard": name, "grid_cap": os.environ.get("DTF_STREAM_GRID_CAP", "2048"),
                      "sgd": [round(sgd, 1), round(n * 20 / sgd / 1e6, 2)], "adam": [round(adam, 1), round(n * 28 / adam / 1e6, 2)]}), flush=True)
    del sets
    torch.cuda.empty_cache()
