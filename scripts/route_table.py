"""Which kernel computes which convolution product?  One forward + backward of a model at a batch size with the route
trace of ops/conv.py on; prints one line per distinct (product, layer shape, input size) with its kernel and call count.
usage: route_table.py [model] [batch]        (resnet50 256 by default; inception_v3, resnet101, ...)"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen as miopen_cache
miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import get_model_module
from distributed_tensorflow_amd.ops import conv


def route_table(name="resnet50", batch=256, device="cuda:0", **build_kw):
    """-> ``{(product, (cin, cout, kh, kw, stride), (N, H, W), choice): calls}`` of one training step."""
    dev = torch.device(device)
    mod = get_model_module(name)
    torch.manual_seed(0)
    m = mod.prepare(mod.build(**build_kw), dev).train()
    b = mod.synthetic_batch(batch, device=dev, seed=1)
    with conv.route_trace() as log:
        mod.loss_fn(m, b).backward()
        torch.cuda.synchronize()
    table = collections.OrderedDict()
    for product, layer, size, choice in log:
        k = (product, layer, size, choice)      # one shape can be met with and without the statistics epilogue
        table[k] = table.get(k, 0) + 1
    return table


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    kw = {}
    if name.startswith("resnet") and name != "resnet50":
        kw["depth"], name = int(name[6:]), "resnet50"
    t = route_table(name, batch, **kw)
    print("%s batch %d, DTF_DETERMINISTIC=%s: %d distinct convolution products" % (sys.argv[1] if len(sys.argv) > 1 else name, batch,
                                                                                   os.environ.get("DTF_DETERMINISTIC", "1"), len(t)))
    print("%-36s %-26s %-16s %5s  %s" % ("product", "cin cout kh kw stride", "N H W", "calls", "kernel"))
    for (product, layer, size, choice), n in sorted(t.items(), key=lambda kv: (-kv[0][2][1], kv[0][1], kv[0][0], kv[0][3])):
        print("%-36s %-26s %-16s %5d  %s" % (product, "%d %d %d %d /%d" % layer, "%d %d %d" % size, n, choice))
    by = collections.Counter()
    for (product, _, _, choice), n in t.items():
        by[choice.split(" + BN")[0].split(",")[0]] += n
    print("\ncalls per kernel family: " + "; ".join("%s %d" % kv for kv in by.most_common()))


if __name__ == "__main__":
    main()
