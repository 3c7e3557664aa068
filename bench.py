#!/usr/bin/env python3
"""Headline benchmark: ResNet-50 synthetic-ImageNet training images/s through the
sharded parameter-server data plane (BASELINE.json metric / config #2).

    python bench.py                                  # 1 GPU, short defaults
    python bench.py --gpus 1 --steps 30 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 30 --warmup 10

Every step is a full training step: forward + backward in bf16 NHWC (MIOpen
convolutions, K7 fused BN/add/ReLU epilogues), K5 bucket pack, RCCL
reduce-scatter push, K1 fused SGD-momentum shard apply on fp32 master weights,
RCCL all-gather pull.  Synthetic images/labels, random-init weights.  Weak
scaling: batch 256 per GPU whatever N.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# upstream tf_cnn_benchmarks parameter_server rows of BASELINE.md (DGX-1 P100, batch 64/GPU, fp32)
BASELINES = {
    "resnet50": {1: 219.0, 2: 422.0, 4: 852.0, 8: 1734.0},
    "inception_v3": {1: 142.0, 2: 284.0, 4: 569.0, 8: 1131.0},
    "resnet152": {1: 91.8, 2: 181.0, 4: 356.0, 8: 716.0},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0: 256 for resnet, model default otherwise)")
    ap.add_argument("--model-arg", action="append", default=[], help="key=value forwarded to the model's build()")
    ap.add_argument("--image-size", type=int, default=0, help="0: model default (224 / 299)")
    ap.add_argument("--update-mode", default="sync", choices=["sync", "stale", "async"],
                    help="async: one-sided applies over peer-mapped PS shards, no collective in the step")
    ap.add_argument("--staleness", type=int, default=1, help="stale mode: updates become visible this many steps late")
    ap.add_argument("--wire-dtype", default="auto", choices=["auto", "bf16", "fp32"],
                    help="dtype of the gradient / parameter traffic (auto: bf16 on GPU, fp32 on CPU)")
    ap.add_argument("--num-ps", type=int, default=0, help="shard groups (detail.ps); 0 or >= N: one per GPU")
    ap.add_argument("--device", default=os.environ.get("DTF_BENCH_DEVICE", "cuda"), choices=["cuda", "cpu"],
                    help="cpu: gloo rehearsal of the multi-rank code path (fp32, use a small --batch/--image-size)")
    ap.add_argument("--push", default="reduce_scatter", choices=["reduce_scatter", "all_to_all"])
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    ap.add_argument("--miopen-find", type=int, default=int(os.environ.get("DTF_MIOPEN_FIND", "0")),
                    help="1: let MIOpen benchmark each conv during warm-up (cudnn.benchmark)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("DTF_BENCH_GRAPH", "0")),
                    help="0 (default): eager -- the step is GPU-bound and a replayed hipGraph runs its nodes without the "
                         "compute / weight-gradient / comm stream overlap the eager queue gets (measured 20.6 vs 19.9 "
                         "ms per step on one MI355X); 1: capture the whole step (forward, backward, hook-launched PS "
                         "pipeline incl. its RCCL collectives) in one hipGraph and replay it, any N, with automatic "
                         "fallback to eager when the capture fails on any rank (host time 2 ms instead of 15 ms per "
                         "step); -1: capture on one GPU, then keep whichever a short timed trial finds faster")
    ap.add_argument("--lr-warmup-steps", type=int, default=100,
                    help="gradual warm-up (Goyal et al.): lr = peak * min(1, step / N); 0: constant peak lr")
    ap.add_argument("--profile-dir", default="")
    a = ap.parse_args()

    # stdout carries exactly ONE line, the result JSON.  Libraries write there too (RCCL prints a version banner
    # on communicator creation), so file descriptor 1 is pointed at stderr for the run and the JSON goes to the
    # saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("DTF_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["DTF_BENCH_WATCHDOG"]), exit=True)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import distributed_tensorflow_amd as _pkg

    if a.graph > 0:
        # the package asked for 8 hardware queues on import (the eager step's streams need them); a replayed graph
        # wants HIP's default back -- see restore_hw_queues().  --graph -1 (trial) keeps 8: eager is the likely winner
        _pkg.restore_hw_queues()
    from distributed_tensorflow_amd.utils import miopen as miopen_cache

    miopen_db = miopen_cache.setup()      # shipped find-db + kernel cache (offline-tuned convs)
    from distributed_tensorflow_amd.utils import gemm_tuning

    gemm_db = gemm_tuning.setup()          # shipped hipBLASLt / rocBLAS solution choices (read-only, no run-time tuning)
    import torch
    import torch.distributed as dist

    gemm_tuning.after_torch_import()

    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.ops import hip
    from distributed_tensorflow_amd.parallel.comm import Comm
    from distributed_tensorflow_amd.parallel.ps_runtime import ShardedPS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            print("bench.py: --gpus %d needs torchrun (WORLD_SIZE=%d)" % (a.gpus, world), file=sys.stderr)
            sys.exit(2)
    on_gpu = a.device == "cuda"
    if on_gpu:
        assert torch.cuda.is_available(), "bench.py needs an MI355X (or --device cpu for the gloo rehearsal)"
        hip.lib()
    # DTF_FORCE_COMM=1: keep a live 1-rank RCCL group so the collectives really run on one GPU
    comm = Comm.from_env(device=None if on_gpu else torch.device("cpu"),
                         force=bool(int(os.environ.get("DTF_FORCE_COMM", "0"))),
                         backend=os.environ.get("DTF_COMM_BACKEND") or (None if on_gpu else "gloo"))
    dev = comm.device

    def dev_sync():
        if on_gpu:
            torch.cuda.synchronize()
    torch.backends.cudnn.benchmark = bool(a.miopen_find)
    torch.manual_seed(1234)

    key = a.model.lower().replace("-", "_")
    mod = get_model_module(key)
    kw = dict(kv.split("=", 1) for kv in a.model_arg)
    if key.startswith("resnet") and key != "resnet50":
        kw["depth"] = int(key[6:])
    if getattr(mod, "NEEDS_COMM", False):
        kw.update(comm=comm, device=dev)
    if not a.batch:
        a.batch = {"inception_v3": 256, "inception3": 256, "bert": 64, "bert_base": 64, "wide_deep": 65536,
                   "mnist_mlp": 256, "mnist": 256}.get(key, 256)
    unit = {"bert": "sequences/sec", "bert_base": "sequences/sec", "wide_deep": "examples/sec"}.get(key, "images/sec")
    model = mod.build(**kw)
    model = mod.prepare(model, dev) if hasattr(mod, "prepare") else model.to(dev)
    model.train()
    opt = mod.default_optimizer()
    wire = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(a.wire_dtype)
    if a.update_mode == "async":
        from distributed_tensorflow_amd.parallel.async_ps import AsyncPS

        ps = AsyncPS(list(model.named_parameters()), comm, opt, bucket_mb=a.bucket_mb, wire_dtype=wire)
    else:
        ps = ShardedPS(list(model.named_parameters()), comm, opt, bucket_mb=a.bucket_mb,
                       update_mode=a.update_mode, push=a.push, num_shards=a.num_ps or None,
                       staleness=a.staleness, wire_dtype=wire)
    if a.update_mode == "sync" and hasattr(ps, "attach"):
        ps.attach(model)        # forward-order pull prefetch (eager steps at N > 1; a no-op on one rank)
    bkw = {"image_size": a.image_size} if a.image_size else {}
    if getattr(mod, "NEEDS_COMM", False):
        bkw["model"] = model
    if "seq_len" in kw:
        bkw["seq_len"] = int(kw["seq_len"])
    batch = mod.synthetic_batch(a.batch, device=dev, seed=100 + comm.rank, **bkw)
    graph_auto = a.graph < 0
    if graph_auto:
        a.graph = 1 if (on_gpu and comm.world_size == 1 and a.update_mode != "async") else 0

    # lr schedule: the learning rate lives in device memory (ps.hyper), so a captured graph replays with new values
    peak_lr, sched = ps.lr, {"t": 0}

    def next_lr():
        sched["t"] += 1
        if a.lr_warmup_steps > 0:
            ps.set_lr(peak_lr * min(1.0, sched["t"] / float(a.lr_warmup_steps)))

    def raw_step():
        loss = mod.loss_fn(model, batch)
        loss.backward()
        ps.step()
        return loss

    def step():
        next_lr()
        return raw_step()

    first_loss = None

    graph = None
    if a.graph and on_gpu:
        # Whole-step hipGraph: forward + backward + the PS pipeline (bucket packs, RCCL collectives on the comm
        # stream forked from the backward hooks, shard applies) become ONE graph launch per step -- the ~600 kernel
        # launches of a step cost the host 18 ms otherwise.  Warm up eagerly on a side stream first (MIOpen kernel
        # selection, allocator, RCCL channel setup must not happen inside the capture), then capture one step.
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(3, a.warmup)):
                lw = step()
                if first_loss is None:
                    first_loss = float(lw.detach().float().item())
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        comm.barrier()
        torch.cuda.synchronize()
        # RCCL's watchdog thread polls the events of the warm-up collectives; an event query from ANY thread is
        # illegal while a global-mode capture is open.  Let it retire them, and capture in thread-local mode.
        time.sleep(0.5)
        ok = torch.ones(1, device=dev)
        if hasattr(ps, "disable_prefetch"):
            ps.disable_prefetch()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                static_loss = raw_step()       # without the lr update: its fill would be replayed as a constant
            graph = g
        except Exception as e:       # noqa: BLE001 -- any capture failure means "run eagerly", never a dead bench
            print("[bench] rank %d: hipGraph capture failed (%s: %s); running eagerly" % (
                comm.rank, type(e).__name__, str(e).splitlines()[0] if str(e) else ""), file=sys.stderr, flush=True)
            ok.zero_()
            torch.cuda.synchronize()
        if comm.world_size > 1:
            comm.all_reduce(ok, op=dist.ReduceOp.MIN)      # every rank replays, or none does
        if float(ok.item()) < 1.0:
            graph = None
        graph_trial = None
        if graph is not None:
            eager_step = step

            def graph_step():
                next_lr()
                if a.lr_warmup_steps > 0 and getattr(ps, "comm_stream", None) is not None:
                    # the rate is filled on the comm stream (after everything launched so far); the replay comes after it
                    torch.cuda.current_stream().wait_stream(ps.comm_stream)
                graph.replay()
                return static_loss

            if graph_auto:
                # which is faster HERE?  A replay costs the host one call instead of ~550 launches, but the step is
                # GPU-bound on one MI355X and the graph's per-node dispatch measured 0.5-1.5 % SLOWER than a full
                # eager queue (profiles/r3_graph_vs_eager.txt).  Six timed steps each way, untimed for the result.
                def trial(fn, n=8):
                    # device-side time of n steps AFTER the queue has filled (three untimed steps first): the host
                    # enqueues an eager step in 13-17 ms against ~20 ms of GPU work, so wall time over a few steps
                    # from an empty queue would charge the eager loop its ramp-up
                    for _ in range(3):
                        fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / n / 1e3

                # alternate and keep the better of two: the first eager trial after the capture also pays for the
                # allocator (the warm-up ran on a side stream, the blocks it cached belong to that stream)
                te, tg = trial(eager_step), trial(graph_step)
                te, tg = min(te, trial(eager_step)), min(tg, trial(graph_step))
                graph_trial = {"eager_ms": round(te * 1e3, 3), "graph_ms": round(tg * 1e3, 3)}
                if tg > te * 0.998:
                    graph = None            # eager is at least as fast: stay eager
            if graph is not None:
                step = graph_step  # noqa: F811
    t_w = time.perf_counter()
    for i in range(a.warmup):
        loss = step()
        if first_loss is None:
            first_loss = float(loss.detach().float().item())
        if comm.rank == 0 and (i < 3 or a.miopen_find):
            dev_sync()
            print("[bench] warmup step %d done at +%.1f s" % (i + 1, time.perf_counter() - t_w), file=sys.stderr,
                  flush=True)
    comm.barrier()
    dev_sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    t_host = time.perf_counter() - t0          # time to ENQUEUE the steps (host-side cost)
    dev_sync()
    comm.barrier()
    dev_sync()
    dt = time.perf_counter() - t0
    # host cost of one step with an EMPTY device queue (outside the timed region): in the timed loop the enqueue of
    # a graph replay or of a GPU-bound eager step also contains queue back-pressure
    probe = []
    for _ in range(3):
        dev_sync()
        tp = time.perf_counter()
        step()
        probe.append(time.perf_counter() - tp)
    dev_sync()
    t_probe = sorted(probe)[1]
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if comm.world_size > 1:
        comm.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    last_loss = float(loss.detach().float().item())
    n = comm.world_size
    global_batch = a.batch * n
    value = global_batch * a.steps / dt
    base = BASELINES.get(key, {}).get(n)
    if comm.rank == 0:
        out = {
            "metric": "%s (whole node) %s synthetic, PS-sharded variables" % (unit, a.model),
            "value": round(value, 2), "unit": unit, "n_gpus": n, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None if base is None else round(value / base, 3), "dtype": "bf16" if on_gpu else "fp32-cpu-rehearsal",
            "data": "synthetic inputs+labels, random-init weights",
            "config": {"model": a.model + ("_v1.5" if key == "resnet50" else ""), "global_batch": global_batch,
                       "per_gpu_batch": a.batch, "image_size": a.image_size or ("299" if "inception" in key else 224),
                       "parallelism": "dp%d+ps-shard%d" % (n, ps.num_shards), "push": ps.push,
                       "pull": "all_gather" if ps.family == "scatter" else "broadcast", "staleness": ps.staleness,
                       "wire_dtype": str(ps.wire_dtype).replace("torch.", ""),
                       "update_mode": a.update_mode, "optimizer": opt.kind, "bucket_mb": a.bucket_mb,
                       "comm_backend": str(comm.backend),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "hipgraph": bool(graph is not None), "hipgraph_trial": graph_trial if a.graph and on_gpu else None, "miopen_find": bool(a.miopen_find), "miopen_shipped_db": bool(miopen_db), "gemm_shipped_db": bool(gemm_db),
                       "baseline": "tf_cnn_benchmarks PS, DGX-1 P100, %s GPU(s): %s img/s" % (n, base),
                       "host_enqueue_ms_per_step": round(t_probe * 1e3, 3),
                       "host_loop_ms_per_step": round(t_host / a.steps * 1e3, 3),
                       "lr_warmup_steps": a.lr_warmup_steps, "lr_now": round(ps.lr, 5),
                       "first_loss": None if first_loss is None else round(first_loss, 4),
                       "final_loss": round(last_loss, 4)},
        }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if a.update_mode == "async":
        ps.close()
    comm.close()


if __name__ == "__main__":
    main()
