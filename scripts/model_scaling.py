#!/usr/bin/env python3
"""MODELLED (not measured) scaling of the sharded-PS step to 2/4/8 MI355X, from the real variable layout, measured
single-GPU kernel numbers and the xGMI link rate ("report measured 1-GPU plus modelled scaling
from measured per-kernel and per-link numbers if 8 GPUs are never available").

    python scripts/model_scaling.py > profiles/r3_modelled_scaling.txt

Model.  The step of rank r is ``forward | backward`` on the compute stream with, per bucket b (reverse layer order),
``K5 pack -> push -> K1/K2 shard apply -> pull`` on the comm stream, started by the autograd hook of the bucket's
last gradient.  A bucket's pipeline hides behind the backward work still to run when it starts; what does not fit
is exposed.  The tail bucket (first layers, 4 MB cap) starts when backward ends: it is exposed in full.

  push / pull of B bytes over N ranks, link rate R (one xGMI link, one direction), efficiency e, latency L:
    ring   : t = L + (N-1)/N * B / (e R)          one link per rank busy, N-1 sequential hops
    direct : t = L + (1/N)   * B / (e R)          slice j straight to rank j, N-1 links at once (all_to_all push;
                                                  what RCCL's all-gather / reduce-scatter do on a full mesh IF it
                                                  picks a direct algorithm -- unknown until measured)
  shard apply: (bytes touched per element) * B/N elements / measured HBM rate of K1 / K2
  compute slow-down while a collective runs: the RCCL kernels occupy c of 256 CUs -> backward work done in that
    window takes 1 / (1 - c/256) longer.
  synchronisation: every collective waits for the slowest rank; a per-step jitter term j (host enqueue is 14-16 ms
    of a 20.6 ms step, so a late host shows up as a late collective) is added once per step.

Inputs marked [measured] come from this repo's profiles; [assumed] are knobs, printed with the table.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LINK_GBS = 153.0          # [spec] one xGMI link, one direction
LAT_US = 25.0             # [assumed] launch + rendezvous latency of one RCCL collective
COMM_CUS = 32             # [assumed] CUs the RCCL kernels occupy while a collective runs
JITTER_MS = {1: 0.0, 2: 0.15, 4: 0.25, 8: 0.4}      # [assumed] slowest-of-N host jitter per step


def layout_of(name, world):
    import torch

    from distributed_tensorflow_amd.models import get_model_module
    from distributed_tensorflow_amd.parallel.layout import FlatLayout
    from distributed_tensorflow_amd.parallel.ps_runtime import trainable

    mod = get_model_module(name)
    with torch.device("meta"):
        model = mod.build()
    opt = mod.default_optimizer()
    named = trainable(model.named_parameters())
    decays = [opt.decays(n) for n, _ in named]
    split = opt.weight_decay > 0 and not all(decays)
    L = FlatLayout([(n, p.numel(), 0 if (d or not split) else 1) for (n, p), d in zip(named, decays)], world,
                   int(32 * (1 << 20) / 2), int(4 * (1 << 20) / 2))
    return L, opt


def collective_us(nbytes, n, eff, direct):
    if n == 1:
        return 0.0
    frac = (1.0 / n) if direct else (n - 1.0) / n
    return LAT_US + frac * nbytes / (eff * LINK_GBS * 1e9) * 1e6


def model(name, step_ms, bwd_frac, apply_tbs, state_bytes_per_elem, clipped, pack_us_total, eff, direct, n):
    L, opt = layout_of(name, n)
    total = float(sum(b.size for b in L.buckets))
    bwd_ms = step_ms * bwd_frac
    # bucket b becomes ready when backward has produced the gradients of everything packed before it: gradients
    # arrive in reverse layer order, time ~ share of the parameters' COMPUTE, approximated by position in backward
    # [assumed]: uniform in the number of variables (ResNet's late layers are parameter-heavy but not time-heavy)
    nvars = float(sum(len(b.params) for b in L.buckets))
    done, exposed_us, busy_until, rows, comm_us_total = 0.0, 0.0, 0.0, [], 0.0
    pend_after = []
    for b in L.buckets:
        done += len(b.params)
        ready_us = bwd_ms * 1e3 * done / nvars
        nbytes = b.size * 2.0
        pack = pack_us_total * b.size / total
        push = collective_us(nbytes, n, eff, direct)
        apply_ = (b.size / n) * state_bytes_per_elem / (apply_tbs * 1e12) * 1e6 + 4.0
        pull = collective_us(nbytes, n, eff, direct)
        if clipped:
            # packs + pushes from the hooks; sumsq -> all-reduce -> apply -> pull of EVERY bucket after backward
            start = max(ready_us, busy_until)
            busy_until = start + pack + push
            pend_after.append((apply_, pull))
        else:
            start = max(ready_us, busy_until)
            busy_until = start + pack + push + apply_ + pull
        comm_us_total += push + pull
        rows.append((b.index, len(b.params), nbytes / 1e6, ready_us / 1e3, pack, push, apply_, pull))
    end_bwd = bwd_ms * 1e3
    if clipped:
        busy_until = max(busy_until, end_bwd) + 30.0 + LAT_US      # sumsq passes + scalar all-reduce
        for apply_, pull in pend_after:
            busy_until += apply_ + pull
    exposed_us = max(0.0, busy_until - end_bwd)
    slow_ms = (comm_us_total - exposed_us if comm_us_total > exposed_us else 0.0) / 1e3 * (1.0 / (1 - COMM_CUS / 256.0) - 1)
    t = step_ms + exposed_us / 1e3 + slow_ms + JITTER_MS[n]
    return t, exposed_us, slow_ms, rows


def table(title, name, step_ms, items_per_step, unit, **kw):
    print("\n== %s  [MODELLED; N=1 measured: %.2f ms/step]" % (title, step_ms))
    print("   %-28s %8s %8s %8s %8s" % ("algorithm, link efficiency", "N=1", "N=2", "N=4", "N=8"))
    for direct in (False, True):
        for eff in (0.5, 0.8):
            cells = []
            for n in (1, 2, 4, 8):
                t, exp_us, slow, _ = model(name, step_ms, eff=eff, direct=direct, n=n, **kw)
                cells.append("%.0f" % (items_per_step * n / t * 1e3))
            effs = []
            for n in (2, 4, 8):
                t, _, _, _ = model(name, step_ms, eff=eff, direct=direct, n=n, **kw)
                effs.append("%.3f" % (step_ms / t))
            print("   %-28s %8s %8s %8s %8s   %s   efficiency vs N=1: %s" % (
                ("direct" if direct else "ring") + ", e = %.1f" % eff, *cells, unit, " / ".join(effs)))
    t, exp_us, slow, rows = model(name, step_ms, eff=0.5, direct=False, n=8, **kw)
    print("   worst case above (ring, e = 0.5, N = 8): exposed after backward %.0f us, compute slow-down %.2f ms, jitter %.2f ms"
          % (exp_us, slow, JITTER_MS[8]))
    print("   buckets at N = 8 (pack order = reverse layer order):")
    print("   %3s %5s %9s %10s %8s %8s %8s %8s" % ("b", "vars", "MB(bf16)", "ready ms", "pack us", "push us", "apply us", "pull us"))
    for r in rows:
        print("   %3d %5d %9.2f %10.2f %8.1f %8.1f %8.1f %8.1f" % r)


def main():
    step_rn = float(os.environ.get("DTF_MODEL_RESNET_MS", "20.63"))     # [measured] BENCH_r02.json (driver-run)
    step_bert = float(os.environ.get("DTF_MODEL_BERT_MS", "26.9"))      # [measured] profiles/r2_all_models.txt: 2380 seq/s @ b64
    print(__doc__.split("Model.")[0].strip())
    print("\nInputs: link %.0f GB/s [spec]; collective latency %.0f us [assumed]; RCCL CUs %d [assumed]; jitter %s ms [assumed];"
          % (LINK_GBS, LAT_US, COMM_CUS, JITTER_MS))
    print("        K5 pack 45 us / step [measured, profiles/r2_resnet50_step.txt]; K1 5.3 TB/s, K2 5.5 TB/s [measured PMC,")
    print("        profiles/r2_pmc_hand_written_kernels.txt]; backward = 2/3 of the step [measured split, r2_host_overhead].")
    table("ResNet-50 v1.5, batch 256/GPU, bf16 wire, SGD-momentum, sync (the headline config)", "resnet50", step_rn, 256,
          "img/s", bwd_frac=0.66, apply_tbs=5.3, state_bytes_per_elem=2 + 4 + 4 + 4 + 4 + 2, clipped=False,
          pack_us_total=45.0)
    table("BERT-base, batch 64/GPU, seq 512, AdamW + global-norm clip (apply + pull of every bucket after backward)",
          "bert", step_bert, 64, "seq/s", bwd_frac=0.66, apply_tbs=5.5,
          state_bytes_per_elem=2 + 4 + 4 + 4 + 4 + 4 + 4 + 2, clipped=True, pack_us_total=190.0)
    print("\nReading: ResNet-50's 51 MB of bf16 gradients are < 0.3 ms of link time per step even on a ring at half the link")
    print("rate; only the 4 MB tail bucket is exposed.  BERT's clipped path exposes the whole 220 MB all-gather on a ring")
    print("(1.3-2.6 ms of a 27 ms step) -- the reason for the per-bucket apply+pull interleave and the forward-order pull")
    print("prefetch in parallel/ps_runtime.py.  None of this replaces a measurement: scripts/scale_sweep.sh produces the")
    print("measured table on a multi-GPU node, tests/test_multigpu.py validates the paths there.")


if __name__ == "__main__":
    main()
