#!/usr/bin/env python3
"""Where does the HOST time of a training step go?  One step at a time with an empty GPU queue: eager enqueue time,
hipGraph replay call time, and a cProfile of the eager step (top entries by own time)."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    from distributed_tensorflow_amd.utils import miopen as miopen_cache

    miopen_cache.setup()
    import torch

    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    dev = torch.device("cuda", 0)
    comm = Comm.from_env(force=bool(int(os.environ.get("DTF_FORCE_COMM", "0"))))
    mod = get_model_module(a.model)
    torch.manual_seed(0)
    model = mod.prepare(mod.build(), dev)
    model.train()
    ps = ShardedPS(list(model.named_parameters()), comm, mod.default_optimizer())
    batch = mod.synthetic_batch(a.batch, device=dev, seed=1)

    def step():
        loss = mod.loss_fn(model, batch)
        loss.backward()
        ps.step()
        return loss

    for _ in range(6):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    print("eager: host enqueue %.2f ms (min %.2f), step wall %.2f ms" % (
        1e3 * sorted(t[0] for t in ts)[len(ts) // 2], 1e3 * min(t[0] for t in ts),
        1e3 * sorted(t[1] for t in ts)[len(ts) // 2]), flush=True)
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(a.top)
    print(s.getvalue()[:9000], flush=True)
    if a.graph:
        time.sleep(0.5)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            step()
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t0))
        print("graph: replay call %.2f ms (min %.2f), step wall %.2f ms" % (
            1e3 * sorted(t[0] for t in ts)[len(ts) // 2], 1e3 * min(t[0] for t in ts),
            1e3 * sorted(t[1] for t in ts)[len(ts) // 2]), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
