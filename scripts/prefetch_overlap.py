#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of an eager run through a live RCCL group: how much of the collectives' time runs
UNDER compute kernels (overlapped) and how much with the compute queue empty (exposed).

    python scripts/prefetch_overlap.py <trace_kernel_trace.csv> [label]

Used to compare DTF_PULL_PREFETCH=1 (per-bucket waits in forward order, parallel/ps_runtime.py::attach) with =0
(the next forward joins the whole comm stream): scripts/gpu_r3_e.sh -> profiles/r3_pull_prefetch_overlap.txt."""
import csv
import sys


def main(path, label=""):
    rows = list(csv.DictReader(open(path)))
    ev = []
    for r in rows:
        n = r["Kernel_Name"]
        comm = ("nccl" in n.lower()) or ("rccl" in n.lower())
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), comm, n))
    ev.sort()
    if not ev:
        print("empty trace")
        return
    # steps are delimited by the optimizer's shard-apply kernels; use the last third of the trace (steady state)
    t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 2 // 3
    comp = [(s, e) for s, e, c, _ in ev if not c and e > t_lo]
    comm = [(s, e, n) for s, e, c, n in ev if c and e > t_lo]
    # merge compute intervals
    merged = []
    for s, e in sorted(comp):
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    tot = ov = 0
    import bisect

    starts = [m[0] for m in merged]
    for s, e, _ in comm:
        tot += e - s
        i = max(0, bisect.bisect_right(starts, s) - 1)
        while i < len(merged) and merged[i][0] < e:
            a, b = max(s, merged[i][0]), min(e, merged[i][1])
            if a < b:
                ov += b - a
            i += 1
    span = ev[-1][1] - t_lo
    busy = sum(b - a for a, b in merged)
    names = {}
    for s, e, n in comm:
        k = n.split("(")[0][:70]
        names[k] = names.get(k, 0) + 1
    print("%s: window %.1f ms, compute kernels busy %.1f ms (%.1f %%), %d collective kernels: total %.2f ms, under "
          "compute %.2f ms (%.0f %%), exposed %.2f ms" % (label or path, span / 1e6, busy / 1e6, 100.0 * busy / span,
                                                          len(comm), tot / 1e6, ov / 1e6,
                                                          100.0 * ov / max(tot, 1), (tot - ov) / 1e6))
    for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4]:
        print("     %5d x %s" % (v, k))


if __name__ == "__main__":
    main(*sys.argv[1:3])
