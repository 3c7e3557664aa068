#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (gpurun_out/pmc/*_counter_collection.csv + *_kernel_trace.csv)
into a per-kernel table: time, MFMA busy, LDS conflicts, HBM bytes and rate."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"


def short(n):
    n = n.replace("void ", "")
    for cut in ("(", ):
        if cut in n:
            n = n[:n.index(cut)]
    return n[:60]


ours = ("gemm_bias_act", "act_bwd_bias", "bn_", "sgd_momentum", "adam_apply", "embedding_", "emb_", "bucket_pack",
        "sumsq", "cast_scale", "colsum", "gemm256p", "gemm_tn", "tn_reduce", "bn2_", "ce_", "plan_", "maxpool", "avgpool",
        "attn_", "ln_", "filter_flip", "conv_tab")
vals = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> [values]
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not any(o in k for o in ours):
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(d, "mfma_kernel_trace.csv"))):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(o in k for o in ours):
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def med(x):
    x = sorted(x)
    return x[len(x) // 2] if x else float("nan")


print("%-52s %9s %7s %9s %9s %8s %8s %8s" % ("kernel (median of dispatches, profiled clocks)", "time_us", "MFMA%", "LDSconf%",
                                             "HBM_MB", "TB/s", "L2hit%", "waves"))
for k in sorted(vals):
    c = vals[k]
    t = med(dur[k])
    gui = med(c.get("GRBM_GUI_ACTIVE", []))
    busy = med(c.get("SQ_VALU_MFMA_BUSY_CYCLES", []))
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
    mf = 100.0 * busy / (gui / 8 * 256 * 4) if gui == gui and gui > 0 and busy == busy else float("nan")
    lc, la = med(c.get("SQ_LDS_BANK_CONFLICT", [])), med(c.get("SQ_LDS_IDX_ACTIVE", []))
    lds = 100.0 * lc / la if la == la and la > 0 else float("nan")
    # FETCH_SIZE under-reads wide streaming loads by exactly 2x on gfx950 (MI355X_MICROARCH.md, HBM)
    fetch_kb, write_kb = med(c.get("FETCH_SIZE", [])), med(c.get("WRITE_SIZE", []))
    mb = (2 * fetch_kb + write_kb) / 1024 if fetch_kb == fetch_kb and write_kb == write_kb else float("nan")
    hit, miss = med(c.get("TCC_HIT_sum", [])), med(c.get("TCC_MISS_sum", []))
    l2 = 100.0 * hit / (hit + miss) if hit == hit and hit + miss > 0 else float("nan")
    print("%-52s %9.1f %7.1f %9.2f %9.1f %8.2f %8.1f %8.0f" % (k, t, mf, lds, mb, mb / t if t == t and t > 0 else float("nan"),
                                                             l2, med(c.get("SQ_WAVES", []))))
print("\nnotes: HBM_MB = (2 x FETCH_SIZE + WRITE_SIZE) KB / 1024 -- FETCH_SIZE counts 128-B requests as 64 B on gfx950;")
print("TB/s = HBM_MB / time (MB/us); MFMA% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs);")
print("LDSconf% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.")
