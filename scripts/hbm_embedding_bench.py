#!/usr/bin/env python3
"""Config #3 at HBM scale: Wide&Deep with 100+ GB of embedding table + optimizer state on ONE MI355X (288 GB HBM3E),
power-law ids.  Reports the end-to-end step and the K3 gather / K4 sparse-apply kernels alone in the HBM random-row
regime (tables >> 256 MiB Infinity Cache): rows/s and effective TB/s (bytes the kernel must move / time)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-table", type=float, default=1.2e7)
    ap.add_argument("--features", type=int, default=26)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--optimizer", default="adagrad")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--skew", default="cubic", choices=["cubic", "zipf", "uniform"])
    ap.add_argument("--out", default="gpurun_out/hbm_embedding.json")
    a = ap.parse_args()
    import numpy as np
    import torch

    from distributed_tensorflow_amd.models import wide_deep as wd
    from distributed_tensorflow_amd.ops import embedding as E
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    dev = torch.device("cuda", 0)
    comm = Comm.from_env(force=bool(int(os.environ.get("DTF_FORCE_COMM", "0"))))
    rows = int(a.rows_per_table)
    t0 = time.time()
    model = wd.build(comm=comm, device=dev, num_features=a.features, rows_per_table=rows, embedding_dim=a.dim,
                     emb_optimizer=a.optimizer)
    model = wd.prepare(model, dev)
    torch.cuda.synchronize()
    emb = model.embeddings
    tbytes = emb.table.numel() * emb.table.element_size()
    abytes = 0 if emb.accumulator is None else emb.accumulator.numel() * 4
    free, total = torch.cuda.mem_get_info()
    print("[hbm] tables %.1f GB + accumulator %.1f GB built in %.1f s; HBM used %.1f of %.1f GB" % (
        tbytes / 1e9, abytes / 1e9, time.time() - t0, (total - free) / 1e9, total / 1e9), flush=True)
    ps = ShardedPS(list(model.named_parameters()), comm, wd.default_optimizer())
    g = torch.Generator().manual_seed(7)
    B, Fn = a.batch, a.features
    if a.skew == "zipf":
        z = np.random.default_rng(7).zipf(1.05, size=(B, Fn)).astype(np.int64)
        ids = torch.from_numpy((z - 1) % rows)
    elif a.skew == "uniform":
        ids = torch.randint(0, rows, (B, Fn), generator=g)
    else:
        ids = (rows * torch.rand(B, Fn, generator=g).pow(3.0)).long().clamp_(0, rows - 1)
    dense = torch.randn(B, model.dense_features, generator=g)
    y = ((ids[:, 0] % 2 == 0) ^ (dense[:, 0] > 0)).float()
    batch = (ids.to(dev), dense.to(dev), y.to(dev))
    uniq = int(torch.unique(ids + torch.arange(Fn).unsqueeze(0) * rows).numel())

    def step():
        loss = wd.loss_fn(model, batch)
        loss.backward()
        ps.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps

    # the two kernels alone, same ids, in the HBM regime
    flat = (batch[0] + emb.table_offsets.unsqueeze(0)).reshape(-1).contiguous()
    n, D = flat.numel(), emb.dim
    grad = torch.randn(n, D, device=dev, dtype=torch.bfloat16) * 1e-3

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    t_gather = timed(lambda: E.embedding_gather(emb.table, flat, torch.bfloat16))
    t_apply = timed(lambda: E.embedding_sparse_apply(emb.table, emb.accumulator, flat, grad, emb.optimizer, 1e-4,
                                                     1e-10, 1.0, 0.0, 1, False))
    acc_w = 0 if emb.accumulator is None else (D if emb.accumulator.dim() == 2 else 1)
    gather_bytes = n * D * 4 + n * D * 2 + n * 8
    apply_bytes = n * D * 2 + n * 8 + uniq * (2 * D * 4 + 2 * acc_w * 4)
    out = {
        "config": {"features": Fn, "rows_per_table": rows, "dim_incl_wide_pad": D, "batch": B, "skew": a.skew,
                   "optimizer": a.optimizer, "table_GB": round(tbytes / 1e9, 2), "accumulator_GB": round(abytes / 1e9, 2),
                   "hbm_used_GB": round((total - free) / 1e9, 1), "lookups_per_step": n, "unique_rows_per_step": uniq,
                   "live_rccl_group": bool(comm.live)},
        "step_ms": round(dt * 1e3, 3), "examples_per_sec": round(B / dt, 1), "final_loss": float(loss.detach()),
        "k3_gather": {"ms": round(t_gather * 1e3, 4), "rows_per_sec": round(n / t_gather, 0),
                      "effective_TBps": round(gather_bytes / t_gather / 1e12, 3)},
        "k4_sort_segment_apply": {"ms": round(t_apply * 1e3, 4), "rows_per_sec": round(n / t_apply, 0),
                                  "effective_TBps_incl_sort": round(apply_bytes / t_apply / 1e12, 3)},
        "dropped_lookups": int(emb.check_overflow(grow=False)) if comm.live else 0,
    }
    print(json.dumps(out, indent=1), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    comm.close()


if __name__ == "__main__":
    main()
