"""One forward + backward of a bench model at the bench batch size; saves every gradient (fp32) to argv[1].
Compare two runs (e.g. shipped tuning files vs DTF_GEMM_TUNING=0 DTF_MIOPEN_ALL_SOLVERS=1 MIOPEN_USER_DB_PATH=/nonexistent)
with `python scripts/grad_dump.py --compare a.pt b.pt`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--compare":
    import torch
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    worst = []
    for n in a:
        d = float((a[n] - b[n]).norm() / (b[n].norm() + 1e-30))
        worst.append((d, n))
    worst.sort(reverse=True)
    print("loss %.6f vs %.6f; largest relative gradient differences:" % (float(a["__loss__"]), float(b["__loss__"])))
    for d, n in worst[:8]:
        print("  %.3e  %s" % (d, n))
    sys.exit(1 if worst[0][0] > 0.2 else 0)
from distributed_tensorflow_amd.utils import miopen as miopen_cache
if os.environ.get("DTF_NO_MIOPEN_DB") != "1":
    miopen_cache.setup()
from distributed_tensorflow_amd.utils import gemm_tuning
gemm_tuning.setup()
import torch
gemm_tuning.after_torch_import()
from distributed_tensorflow_amd.models import get_model_module

mod = get_model_module(sys.argv[3] if len(sys.argv) > 3 else "resnet50")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = mod.build()
m = (mod.prepare(m, dev) if hasattr(mod, "prepare") else m.to(dev)).train()
batch = mod.synthetic_batch(int(sys.argv[2]) if len(sys.argv) > 2 else 256, device=dev, seed=1)
loss = mod.loss_fn(m, batch)
loss.backward()
torch.cuda.synchronize()
out = {n: p.grad.float().cpu() for n, p in m.named_parameters()}
out["__loss__"] = loss.detach().float().cpu()
torch.save(out, sys.argv[1])
print("saved", sys.argv[1], "loss", float(loss))
