"""Worker / PS task bodies (new layer N3): what the user's TF1 script does inside
the reference's ``tf-dis`` image (external E0) -- build the model,
join the cluster, and loop ``pull -> forward/backward -> push -> apply``.

``run_task`` is called by ``service/entry.py`` in every task process of a job;
``train_loop`` is the reusable loop (also used by ``bench.py``).
"""
from __future__ import annotations

import os
import sys
import time

import torch

from ..models import get_model_module
from ..parallel import checkpoint as ckpt
from ..parallel.comm import Comm
from ..parallel.ps_runtime import OptimizerSpec, ShardedPS


# graph=auto keeps the replay only if it is at least this much faster than the eager step (a factor on the eager time)
GRAPH_AUTO_MARGIN = float(os.environ.get("DTF_GRAPH_AUTO_MARGIN", "0.98"))


def _wire_dtype(opts, device):
    if opts.wire_dtype in ("bf16", "bfloat16"):
        return torch.bfloat16
    if opts.wire_dtype in ("fp32", "float32"):
        return torch.float32
    return torch.bfloat16 if device.type == "cuda" else torch.float32


def _optimizer(mod, opts):
    spec = mod.default_optimizer() if hasattr(mod, "default_optimizer") else OptimizerSpec()
    if opts.optimizer:
        spec.kind = opts.optimizer
    if opts.lr > 0:
        spec.lr = opts.lr
    return spec


def _parse_fault(s):
    """``"exit:worker=1,step=3,code=17"`` -> dict (fault injection)."""
    if not s:
        return None
    kind, _, rest = s.partition(":")
    d = {"kind": kind, "worker": 0, "step": 1, "code": 17}
    for kv in rest.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            d[k] = int(v)
    return d


def _inject(fault, worker, step):
    if fault and fault["worker"] == worker and fault["step"] == step:
        print("[fault] injecting %s at worker %d step %d" % (fault["kind"], worker, step), flush=True)
        if fault["kind"] == "exit":
            os._exit(fault["code"])
        if fault["kind"] == "hang":
            while True:
                time.sleep(3600)
        if fault["kind"] == "raise":
            raise RuntimeError("injected fault")


def unknown_options(mod, extra):
    """Keys of the job's query string that nothing reads: not a TrainOptions field (those never reach ``extra``), not a
    named argument of the model's ``build``, not one of its ``OPTIONS``, not a synthetic-batch shape."""
    import inspect

    known = {"seq_len", "image_size"} | set(getattr(mod, "OPTIONS", ()))
    known |= {n for n, p in inspect.signature(mod.build).parameters.items() if p.kind != p.VAR_KEYWORD}
    return sorted(k for k in extra if k not in known), sorted(known - {"comm", "device"})


def prepare_model(mod, opts, device, comm=None):
    kw = dict(opts.extra)
    bad, known = unknown_options(mod, kw)
    if bad:
        # a typo ("step=100", "batchsize=512") would otherwise run the default silently
        print("[train] WARNING: option(s) %s are not read by anything and were ignored (model options: %s; job options: "
              "docs/SWITCHES.md)" % (", ".join(bad), ", ".join(known)), flush=True)
    if getattr(mod, "NEEDS_COMM", False):
        kw.update(comm=comm, device=device)
    model = mod.build(**kw)
    if hasattr(mod, "prepare"):
        model = mod.prepare(model, device)
    else:
        model = model.to(device)
    return model


def _embedding_modules(model):
    from ..parallel.embedding import ShardedEmbedding

    return [(n, m) for n, m in model.named_modules() if isinstance(m, ShardedEmbedding)]


def replicated_state(model, ps, step=None):
    """Non-PS state saved by the chief: global_step and BatchNorm moving statistics (sharded
    embedding tables are NOT replicated state: every rank writes its own part)."""
    skip = tuple(n + "." for n, _ in _embedding_modules(model))
    rep = {"global_step": torch.tensor(ps.global_step if step is None else step, dtype=torch.int64)}
    for n, b in model.named_buffers():
        if not n.startswith(skip) if skip else True:
            rep[n] = b
    return rep


def save_checkpoint(model, ps, export, step, tf_names, chief):
    sections, variables = {}, {}
    for _, m in _embedding_modules(model):
        sections.update(m.checkpoint_sections())
        variables.update(m.checkpoint_variables())
    return ckpt.save_ps(ps, export, step, tf_names, replicated_state(model, ps) if chief else None,
                        extra_sections=sections or None, extra_variables=variables or None)


def restore_checkpoint(model, ps, export, tf_names, step=None):
    """Resume: dense PS state (any world size), replicated buffers, and the row-sharded embedding tables with
    their accumulators (re-partitioned when the world size or the partition strategy changed).  Returns the step
    (``step=None``: the newest complete checkpoint)."""
    rd = ckpt.restore_ps(ps, export, step=step, tf_names=tf_names)
    rep = rd.replicated()
    skip = tuple(n + "." for n, _ in _embedding_modules(model))
    with torch.no_grad():
        for n, b in model.named_buffers():
            if n in rep and not (skip and n.startswith(skip)):
                b.copy_(rep[n])
        for _, m in _embedding_modules(model):
            m.restore(rd)
    return rd.step


def lr_schedule(opts, base_lr, total_steps):
    """``step (0-based, of the whole job) -> learning rate`` from the job options ``lr_warmup_steps``, ``lr_schedule``,
    ``lr_boundaries``, ``lr_decay_factor``, ``lr_final_fraction``; ``None`` when the rate is constant.  What
    tf_cnn_benchmarks-style scripts do with ``tf.train.piecewise_constant`` / a warm-up ramp."""
    import math

    warm = max(0, int(getattr(opts, "lr_warmup_steps", 0) or 0))
    kind = str(getattr(opts, "lr_schedule", "constant") or "constant").lower()
    if kind not in ("constant", "cosine", "piecewise"):
        raise ValueError("lr_schedule must be constant, cosine or piecewise, not %r" % kind)
    if warm == 0 and kind == "constant":
        return None
    total = max(1, int(total_steps))
    bounds = sorted(float(b) for b in str(getattr(opts, "lr_boundaries", "") or "").split(",") if b.strip())
    factor = float(getattr(opts, "lr_decay_factor", 0.1))
    final = float(getattr(opts, "lr_final_fraction", 0.0))

    def at(step):
        if step < warm:
            return base_lr * (step + 1) / warm
        if kind == "cosine":
            x = min(1.0, (step - warm) / max(1, total - 1 - warm))
            return base_lr * (final + (1.0 - final) * 0.5 * (1.0 + math.cos(math.pi * x)))
        if kind == "piecewise":
            return base_lr * factor ** sum(1 for b in bounds if step >= b * total)
        return base_lr

    return at


def train_loop(model, mod, ps, batch, steps, *, worker=0, log_every=10, fault=None, on_step=None,
               items_per_step=None, start_step=0, graph=False, batches=None, on_log=None, lr_fn=None):
    """Runs ``steps`` PS steps; returns (last_loss, seconds, steps_done).  ``graph=True``: after three eager steps
    the whole step (forward, backward, hook-launched PS pipeline incl. its collectives) is captured in one hipGraph
    and replayed -- the ~600 launches of a step then cost the host one call (GPU only).  ``graph="auto"``: capture, then
    time eager training steps against replayed ones (two phases of 8 each, the better counts) and keep the faster (all of them count as training steps; with
    several ranks the decision is taken on the slowest rank's times, so every rank replays or none does).

    ``batches``: an iterator of device batches (``train/data.py``: the job's ``data`` directory through the prefetcher, or
    several synthetic batches); ``batch`` is then the first of them.  With ``graph=True`` every new batch is copied INTO
    the tensors of ``batch`` (a replayed graph reads fixed addresses); eagerly the step just takes the new tensors."""
    dev = ps.device
    t0 = time.time()
    last = None
    loss = None
    cur = {"b": batch}
    static_input = bool(graph) and dev.type == "cuda"
    if static_input and batches is not None:
        # the captured step reads these tensors forever and every new batch is copied into them: they must be nobody
        # else's (a round-robin source would otherwise find its first batch overwritten when it comes round again)
        cur["b"] = tuple(t.clone() if isinstance(t, torch.Tensor) else t for t in batch)

    def eager():
        out = mod.loss_fn(model, cur["b"])
        out.backward()
        ps.step()
        return out

    def next_batch(first):
        if batches is None or first:          # `batch` IS the first element of `batches`
            return
        nb = next(batches)
        if static_input:
            with torch.no_grad():
                for d, s in zip(cur["b"], nb):
                    if isinstance(d, torch.Tensor):
                        d.copy_(s)
        else:
            cur["b"] = nb

    replay, static_loss, trial = None, None, None
    TRIAL = (("graph", 3, False), ("graph", 8, True), ("graph", 8, True))
    PRE = 3 + (8 if graph == "auto" else 0)       # eager steps before the capture; in auto mode the last 8 are timed
    pre_t0, eager_ms = None, None
    if graph == "auto" and steps < PRE + sum(n for _, n, _ in TRIAL) + 8:
        graph, static_input = False, False          # no room for a trial: a capture (0.6 s) would be a guess, stay eager
        cur["b"] = batch
    # the eager steps before a capture run on a side stream (the protocol torch.cuda.graph documents and bench.py follows):
    # captured after warm-up steps on the DEFAULT stream, hipStreamEndCapture crashed the process on ROCm 7.2
    warm = None
    if graph and dev.type == "cuda" and steps > PRE:
        warm = torch.cuda.Stream(dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
    for it in range(start_step, start_step + steps):
        _inject(fault, worker, it + 1)
        if lr_fn is not None:
            ps.set_lr(lr_fn(it))        # a device-side scalar the applies read: also valid for a replayed step
            if replay is not None and getattr(ps, "comm_stream", None) is not None:
                # the fill runs on the comm stream, after everything launched so far (the previous replay included); the
                # replay launched next must in turn come after the fill -- eagerly the pull events order that, a replayed
                # graph knows nothing of the real comm stream
                torch.cuda.current_stream(dev).wait_stream(ps.comm_stream)
        warming = warm is not None and replay is None and graph and it - start_step < PRE
        if warming:
            if graph == "auto" and it - start_step == 3:
                torch.cuda.synchronize(dev)
                pre_t0 = time.perf_counter()
            with torch.cuda.stream(warm):
                next_batch(it == start_step)
                loss = eager()
            if on_step is not None:
                on_step(it + 1, loss)
            continue
        next_batch(it == start_step)
        if graph and dev.type == "cuda" and replay is None and it - start_step == PRE:
            if warm is not None:
                torch.cuda.current_stream(dev).wait_stream(warm)
            torch.cuda.synchronize(dev)
            if pre_t0 is not None:
                # the eager step is timed BEFORE the capture: afterwards it measured 2-5 ms per step slower than in a
                # process that never captured (scripts/train_loop_auto_probe.py), which would bias the choice
                eager_ms = (time.perf_counter() - pre_t0) / (PRE - 3)
            live = getattr(getattr(ps, "comm", None), "live", False)
            if live:
                time.sleep(0.5)         # the RCCL watchdog retires the earlier collectives (no event queries mid-capture)
            if hasattr(ps, "disable_prefetch"):
                ps.disable_prefetch()
            step_before = ps.global_step
            ok = torch.ones(1, device=dev)
            g = None
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    static_loss = eager()
            except Exception as e:      # noqa: BLE001 -- stay eager rather than fail the job
                print("[train] hipGraph capture failed (%s); staying eager" % type(e).__name__, flush=True)
                try:
                    # an exception inside capture_end leaves the capture stream current: back to the default stream
                    torch.cuda.set_stream(torch.cuda.default_stream(dev))
                    ok.zero_()
                    torch.cuda.synchronize(dev)
                except Exception:       # noqa: BLE001
                    raise RuntimeError("the hipGraph capture of the training step failed (%s: %s) and left the device "
                                       "in capture mode; run this job without graph=1" % (
                                           type(e).__name__, str(e).splitlines()[0] if str(e) else "")) from e
            comm = getattr(ps, "comm", None)
            if comm is not None and comm.live and comm.world_size > 1:
                # every rank replays or none does: a rank replaying captured collectives against ranks issuing
                # them eagerly is legal, but a rank whose capture died midway must not be the only eager one
                import torch.distributed as dist

                comm.all_reduce(ok, op=dist.ReduceOp.MIN)
            # the capture (complete or not) ran no kernel but moved the host-side bookkeeping of one step
            if hasattr(ps, "reset_step_state"):
                ps.reset_step_state(step_before)
            else:
                ps.global_step = step_before
            if float(ok.item()) >= 1.0:
                replay = g
                print("[train] step captured in a hipGraph", flush=True)
                if graph == "auto" and eager_ms is not None and start_step + steps - it >= sum(n for _, n, _ in TRIAL) + 8:
                    # which is faster HERE?  The next replays are timed (the first three are not, the better of two
                    # phases counts) and compared with the eager steps timed before the capture
                    trial = {"plan": list(TRIAL), "best": {"eager": eager_ms, "graph": 1e30}}
                    trial["phase"], trial["left"], trial["timed"] = trial["plan"].pop(0)
                    trial["n"], trial["t0"] = trial["left"], time.perf_counter()
            else:
                graph, static_loss = False, None
        if replay is not None and (trial is None or trial["phase"] == "graph"):
            replay.replay()
            ps.global_step += 1
            loss = static_loss
        else:
            loss = eager()
        if trial is not None:
            trial["left"] -= 1
            if trial["left"] == 0:
                torch.cuda.synchronize(dev)
                now = time.perf_counter()
                if trial["timed"]:
                    trial["best"][trial["phase"]] = min(trial["best"][trial["phase"]], (now - trial["t0"]) / trial["n"])
                if trial["plan"]:
                    trial["phase"], trial["left"], trial["timed"] = trial["plan"].pop(0)
                    trial["n"], trial["t0"] = trial["left"], time.perf_counter()
                else:
                    t = torch.tensor([trial["best"]["eager"], trial["best"]["graph"]], dtype=torch.float64, device=dev)
                    comm = getattr(ps, "comm", None)
                    if comm is not None and comm.live and comm.world_size > 1:
                        import torch.distributed as dist

                        comm.all_reduce(t, op=dist.ReduceOp.MAX)        # one decision for the job: the slowest rank's times
                    te, tg = float(t[0]), float(t[1])
                    keep = tg < te * GRAPH_AUTO_MARGIN
                    print("[train] hipGraph trial: eager %.2f ms/step, replay %.2f ms/step -> %s" % (
                        te * 1e3, tg * 1e3, "replaying" if keep else "staying eager"), flush=True)
                    if not keep:
                        # drop the graph and its private memory pool, and give the cached blocks of the warm-up stream
                        # back: the eager loop continues on the default stream
                        replay, static_loss, g = None, None, None
                        loss = loss.detach().clone() if loss is not None else None
                        torch.cuda.synchronize(dev)
                        torch.cuda.empty_cache()
                    trial = None
        if on_step is not None:
            on_step(it + 1, loss)
        if log_every and (it + 1) % log_every == 0:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            now = time.time()
            last = float(loss.detach().float().item())
            rate = (it + 1 - start_step) * (items_per_step or 0) / max(now - t0, 1e-9)
            print("[train] step %d loss %.5f  %.1f items/s" % (it + 1, last, rate), flush=True)
            if on_log is not None:
                on_log(it + 1, last, rate)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    if loss is not None:
        last = float(loss.detach().float().item())
    return last, time.time() - t0, steps


def run_task(job_name, task_index, num_ps, num_workers, placement, opts, data_dir, export_dir, metrics,
             t_start=None, on_checkpoint=None):
    """One task of a job, from rendezvous to teardown.  Whatever way the body ends -- worker done, ps task done
    serving, exception -- the process group is destroyed before the caller gets control back: a task that leaves
    the interpreter with a live gloo / RCCL group aborts in ``std::terminate`` and turns a finished job into a
    failed one (``Comm.close``)."""
    comms = []
    ok = False
    try:
        rc = _run_task(job_name, task_index, num_ps, num_workers, placement, opts, data_dir, export_dir, metrics,
                       t_start, comms, on_checkpoint or (lambda step, comm, chief: None))
        ok = rc == 0
        return rc
    finally:
        for c in comms:
            # success: every rank has passed the final barrier, the destroy is immediate.  Failure: peers may be
            # stuck in a collective this rank will never join -- bounded destroy, then a hard exit with the error
            c.close(timeout_s=None if ok else 15.0, on_timeout_exit=None if ok else 1)


def _run_task(job_name, task_index, num_ps, num_workers, placement, opts, data_dir, export_dir, metrics,
              t_start, comms, on_checkpoint):
    """``on_checkpoint(step, comm, chief)`` runs on every rank once a checkpoint is complete on local disk
    (``service/entry.py`` mirrors it to an ``hdfs://`` export there)."""
    t_start = t_start or time.time()
    trace = os.environ.get("DTF_TRACE_START", "0") == "1"

    def mark(what):
        if trace:
            print("[start +%.3f s] %s" % (time.time() - t_start, what), flush=True)
    from ..utils import gemm_tuning

    gemm_tuning.after_torch_import()      # shipped GEMM choices (if entry.py enabled them) are read-only
    mod = get_model_module(opts.model)
    if int(getattr(opts, "deterministic", -1)) >= 0:
        from ..ops import conv as conv_ops

        conv_ops.set_deterministic(opts.deterministic)
    use_cuda = torch.cuda.is_available() and int(os.environ.get("DTF_NUM_GPUS", "1")) > 0
    device = torch.device("cuda", 0) if use_cuda else torch.device("cpu")
    torch.manual_seed(opts.seed)
    if not use_cuda and num_workers + num_ps > 1 and "OMP_NUM_THREADS" not in os.environ:
        # CPU ranks of one job share the host: without a cap every rank spins cpu_count() threads
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // (num_workers + num_ps)))
    # the colocated asynchronous mode uses NO collective in the step (one-sided applies over peer-mapped shards):
    # its process group is control-plane only, and gloo lets several ranks share a GPU
    one_sided = placement != "dedicated" and opts.update_mode == "async"
    # DTF_FORCE_COMM=1: keep a live process group even with one rank (the collectives then really run: 1-GPU rehearsal)
    comm = Comm.from_env(device, backend="gloo" if one_sided else None,
                         force=bool(int(os.environ.get("DTF_FORCE_COMM", "0"))) and not one_sided)
    comms.append(comm)
    chief = job_name == "worker" and task_index == 0
    mark("communicator ready")
    if chief:
        metrics(t_comm_ready=time.time())
    wire = _wire_dtype(opts, device)
    opt = _optimizer(mod, opts)
    ckpt.KEEP = int(opts.keep_checkpoints)
    export = export_dir if export_dir and export_dir not in ("/tmp", "") else (export_dir or "")
    do_ckpt = bool(export_dir) and export_dir != "/tmp"

    # ---------------------------------------------------------------- PS task
    if placement == "dedicated" and job_name == "ps":
        from ..parallel.dedicated_ps import PSServer

        with torch.device("meta"):
            meta_model = mod.build(**opts.extra)
        from ..parallel.ps_runtime import trainable

        shapes = [(n, p.numel()) for n, p in trainable(meta_model.named_parameters())]
        srv = PSServer(shapes, comm, num_ps, opt, opts.bucket_mb, wire, device,
                       "async" if opts.update_mode in ("async", "stale") else "sync")
        print("[ps %d] serving shard of %d elements (%s)" % (task_index, srv.layout.shard_total,
                                                             srv.update_mode), flush=True)
        from ..parallel.dedicated_ps import agree_on_step

        def ps_sections():
            sec = {"": srv.master}
            names = opt.slots
            if names:
                sec[names[0]] = srv.slot_m
            if len(names) > 1:
                sec[names[1]] = srv.slot_v
            return sec

        def write_shard(step):
            ckpt.write_shard(export, step, task_index, num_ps, ps_sections())
            on_checkpoint(step, None, False)
            print("[ps %d] wrote checkpoint shard of step %d" % (task_index, step), flush=True)

        resumed = False
        if opts.resume and do_ckpt and ckpt.latest_step(export) is not None:
            tfn = mod.tf_variable_names(meta_model) if hasattr(mod, "tf_variable_names") else {}
            rd = srv.restore(export, trainable(meta_model.named_parameters()), tfn)
            resumed = True
            print("[ps %d] resumed shard from step %d" % (task_index, rd.step), flush=True)
        steps = agree_on_step(comm, srv.serve(ckpt_every=opts.ckpt_every if do_ckpt else 0, write_shard=write_shard,
                                              skip_init=resumed))
        print("[ps %d] all workers done after %d applies" % (task_index, steps), flush=True)
        if do_ckpt:
            write_shard(steps)          # the shard is (remotely) in place before the barrier below
        # shards first, index second: the chief publishes the index + 'checkpoint' pointer only after every PS task
        # has passed this barrier, so a reader that follows the pointer always finds the data files
        comm.barrier()
        return 0

    # ------------------------------------------------------------ worker task
    model = prepare_model(mod, opts, device, comm)
    mark("model built on the device")
    named = list(model.named_parameters())
    if placement == "dedicated":
        from ..parallel.dedicated_ps import PSWorkerClient

        ps = PSWorkerClient(named, comm, num_ps, opt, opts.bucket_mb, wire,
                            "async" if opts.update_mode in ("async", "stale") else "sync")
        ps.device = device
    elif one_sided:
        from ..parallel.async_ps import AsyncPS

        ps = AsyncPS(named, comm, opt, opts.bucket_mb, wire)
    else:
        # detail.ps = number of shard groups hosting the variables (the reference creates that many PS tasks,
        # handlers/train_handler.py:96-112); P >= W (or ps_shards=auto) spreads them over every worker rank
        shards = num_ps if (opts.ps_shards == "spec" and num_ps >= 1) else None
        ps = ShardedPS(named, comm, opt, opts.bucket_mb, wire, "stale" if opts.update_mode == "stale" else "sync",
                       opts.push, num_shards=shards, staleness=opts.staleness)
    if isinstance(ps, ShardedPS):
        ps.attach(model)            # forward-order pull prefetch (eager, W > 1)
    print("[worker %d] PS runtime: %s, update_mode=%s, shard groups=%d, push=%s" % (
        task_index, type(ps).__name__, ps.update_mode, getattr(ps, "num_shards", num_ps), getattr(ps, "push", "-")),
        flush=True)
    mark("PS runtime ready")
    tf_names = mod.tf_variable_names(model) if hasattr(mod, "tf_variable_names") else {}
    start_step = 0
    if opts.resume and do_ckpt and ckpt.latest_step(export) is not None:
        if placement == "dedicated":
            # the PS tasks restore master + slots of their shards; a worker restores what it computes with and
            # the replicated state.  Asynchronous checkpoints are labelled in applies (W per worker step).
            rd = ps.restore(export, tf_names)
            rep = rd.replicated()
            with torch.no_grad():
                for n, b in model.named_buffers():
                    if n in rep:
                        b.copy_(rep[n])
            start_step = rd.step if ps.update_mode == "sync" else rd.step // max(1, num_workers)
        else:
            start_step = restore_checkpoint(model, ps, export, tf_names)
        print("[worker %d] resumed from step %d" % (task_index, start_step), flush=True)
    bkw = {k: int(v) for k, v in opts.extra.items() if k in ("seq_len", "image_size")}
    if getattr(mod, "NEEDS_COMM", False):
        bkw["model"] = model
    batches = None
    dataset = None
    if opts.input != "synthetic":
        from . import data as data_mod

        dataset = data_mod.open_dataset(data_dir)
    if dataset is not None:
        # the job's `data` directory (reference: argv[6] of the training script, handlers/train_handler.py:82-83)
        stream = data_mod.BatchStream(dataset, opts.batch, worker=task_index, num_workers=num_workers, seed=opts.seed,
                                      start_step=start_step)
        batches = data_mod.Prefetcher(stream, data_mod.transform_for(mod, dataset, device), device)
        batch = next(batches)
        print("[worker %d] input: %d samples from %s (%s), %d steps per epoch" % (
            task_index, len(dataset), dataset.source, dataset.kind, stream.per_epoch), flush=True)
        mark("first batch of the dataset on the device")
    else:
        batch = mod.synthetic_batch(opts.batch, device=device, seed=opts.seed + 1 + task_index, **bkw)
        if opts.synthetic_batches > 1:
            from . import data as data_mod

            batches = data_mod.Cycle([batch] + [
                mod.synthetic_batch(opts.batch, device=device, seed=opts.seed + 1 + task_index + 7919 * k, **bkw)
                for k in range(1, opts.synthetic_batches)])
            batch = next(batches)
        mark("synthetic batch on the device")
    fault = _parse_fault(opts.fault)
    first = {"done": False}

    def on_step(step, loss):
        if not first["done"]:
            first["done"] = True
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            if chief:
                now = time.time()
                metrics(t_step1=now, first_loss=float(loss.detach().float().item()))
                print("[worker 0] first step done %.3f s after task start" % (now - t_start), flush=True)
        if opts.ckpt_every and do_ckpt and step % opts.ckpt_every == 0:
            if placement != "dedicated":
                save_checkpoint(model, ps, export, step, tf_names, chief)
                on_checkpoint(step, comm, chief)
            elif ps.update_mode == "sync":
                comm.barrier()          # every PS task has written shard `step` (PSServer._serve_sync)
                if chief:
                    publish_dedicated(step)

    def publish_dedicated(step):
        """The chief's part of a dedicated-placement checkpoint: replicated state, then the index (the pointer)."""
        shapes = {nm: tuple(p.shape) for nm, p in ps.named_params}
        orders = {nm: ckpt.storage_order(p) for nm, p in ps.named_params}
        variables = ckpt.dense_variable_index(ps.layout, tf_names, shapes, opt.slots, orders)
        ckpt.write_replicated(export, step, replicated_state(model, ps, step))
        ckpt.write_index(export, step, num_ps, variables, extra={"optimizer": opt.kind, "placement": "dedicated"})
        on_checkpoint(step, None, True)

    remaining = max(0, opts.steps - start_step)
    items = opts.batch * num_workers
    use_graph = opts.graph
    if use_graph == "default":
        use_graph = "auto" if (num_workers == 1 and device.type == "cuda" and placement != "dedicated" and not one_sided
                               and getattr(ps, "update_mode", "") != "stale") else False
    if use_graph:
        # a captured step replays ONE fixed sequence of kernels: a stale pipeline rotates through `staleness` staging
        # slots and waits for events of earlier steps, the one-sided and dedicated runtimes talk to other processes from
        # the host in the middle of a step
        why = ("update_mode=stale" if getattr(ps, "update_mode", "") == "stale" else "update_mode=async" if one_sided
               else "placement=dedicated" if placement == "dedicated" else "")
        if why:
            if chief:
                print("[train] graph=%s ignored: a step with %s cannot be captured in one hipGraph" % (
                    "auto" if use_graph == "auto" else 1, why), flush=True)
            use_graph = False
    on_log = None
    if chief:
        from .summary import ScalarLog

        # the training curve: <export>/scalars.jsonl + a thinned copy in the job's metrics (train/summary.py)
        curve = ScalarLog(export if do_ckpt else None, sink=metrics)
        on_log = lambda step, loss, rate: curve.add(step, loss=loss, items_per_sec=rate,      # noqa: E731
                                                    **({"lr": ps.lr} if hasattr(ps, "lr") else {}))
    try:
        last, secs, n = train_loop(model, mod, ps, batch, remaining, worker=task_index, log_every=opts.log_every,
                                   fault=fault, on_step=on_step, items_per_step=items, start_step=start_step,
                                   graph=use_graph, batches=batches, on_log=on_log,
                                   lr_fn=lr_schedule(opts, float(ps.lr), opts.steps) if (hasattr(ps, "set_lr") and hasattr(ps, "lr")) else None)
    finally:
        if batches is not None:
            batches.close()
    rate = n * items / max(secs, 1e-9)
    print("[worker %d] done: %d steps in %.2f s, %.1f items/s, final loss %s" % (
        task_index, n, secs, rate, last), flush=True)
    if placement == "dedicated":
        applies = ps.finish()
        comm.barrier()          # every PS task has written its shard (see the ps branch above)
        if do_ckpt and chief:
            publish_dedicated(applies)
    else:
        if one_sided:
            ps.finish()         # every worker's last apply has landed
        if do_ckpt:
            save_checkpoint(model, ps, export, ps.global_step, tf_names, chief)
            on_checkpoint(ps.global_step, comm, chief)
        if one_sided:
            ps.close()
    if chief:
        metrics(t_done=time.time(), step=ps.global_step, items_per_sec=rate,
                final_loss=last if last is not None else "nan")
    # last collective of the job: nobody tears the group down while a peer still needs it (the dedicated
    # placement passed its barrier above, after the ps tasks wrote their shards)
    if placement != "dedicated":
        comm.barrier()
    return 0
