#!/usr/bin/env python3
"""Host-side cost of one collective call: torch.distributed (c10d ProcessGroupNCCL = RCCL) vs the native communicator
(native/rccl_comm.cpp).  One rank on one GPU (a live 1-rank communicator), small payload, the stream kept non-empty so
the number is the ENQUEUE cost the autograd-hook thread pays, not RCCL's kernel time.

    python scripts/comm_host_cost.py > gpurun_out/comm_host_cost.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from distributed_tensorflow_amd.parallel.comm import Comm  # noqa: E402
from distributed_tensorflow_amd.parallel.rccl_native import NativeComm  # noqa: E402


def bench(comm, name, n=200):
    dev = comm.device
    full = torch.zeros(1 << 16, device=dev, dtype=torch.bfloat16)
    shard = torch.zeros(1 << 16, device=dev, dtype=torch.bfloat16)
    one = torch.zeros(1, device=dev)
    rows = []
    for what, fn in (("reduce_scatter", lambda: comm.reduce_scatter(shard, full)),
                     ("all_gather", lambda: comm.all_gather(full, shard)),
                     ("all_to_all", lambda: comm.all_to_all(full, shard)),
                     ("all_reduce(1)", lambda: comm.all_reduce(one))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / n
        rows.append((what, t_host * 1e6, t_all * 1e6))
    print("%s" % name)
    for what, h, a in rows:
        print("   %-16s host enqueue %6.1f us / call   (incl. device drain %6.1f us)" % (what, h, a))
    return rows


def main():
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.update(RANK="0", WORLD_SIZE="1")
    c10d = Comm.from_env(dev, backend="nccl", force=True)
    a = bench(c10d, "torch.distributed (ProcessGroupNCCL on RCCL), 1 rank")
    native = NativeComm.from_env(dev)
    b = bench(native, "native communicator (native/rccl_comm.cpp), 1 rank")
    print("ratio c10d / native (host enqueue): " + ", ".join("%s %.1fx" % (x[0], x[1] / y[1]) for x, y in zip(a, b)))
    native.close()
    c10d.close()


if __name__ == "__main__":
    main()
