#!/usr/bin/env python3
"""Which framework (non-own, non-library-GEMM) kernels run in a ResNet-50 step, with input shapes and call sites."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.utils import miopen, gemm_tuning
miopen.setup(); gemm_tuning.setup()
import torch
from torch.profiler import profile, ProfilerActivity
import importlib
name = sys.argv[1] if len(sys.argv) > 1 else "resnet"
resnet = importlib.import_module("distributed_tensorflow_amd.models." + name)
from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS
dev = torch.device("cuda:0")
model = resnet.prepare(resnet.build(), dev)
ps = ShardedPS(list(model.named_parameters()), opt=resnet.default_optimizer())
batch = resnet.synthetic_batch(int(sys.argv[2]) if len(sys.argv) > 2 else 256, device=dev, seed=1)
def step():
    loss = resnet.loss_fn(model, batch); loss.backward(); ps.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
want = ("aten::index_select", "aten::gather", "aten::cat", "aten::index", "aten::embedding", "aten::masked_fill", "aten::where", "aten::div", "aten::sub", "aten::add", "aten::add_", "aten::copy_", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zero_", "aten::mul", "aten::sum", "aten::mean")
rows = []
for e in prof.events():
    if e.name in want and e.device_time_total > 0:
        st = [f for f in (e.stack or []) if "distributed_tensorflow_amd" in f or "bench" in f]
        rows.append((e.device_time_total, e.name, str(e.input_shapes)[:90], st[0][-70:] if st else ""))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%8.1f us  %-16s %-90s %s" % r)
