#!/usr/bin/env python3
"""Per-kernel table from rocprofv3 --pmc passes over WHOLE training steps (scripts/pmc_kernels3.py): one kernel name sees
many shapes in a step, so the columns are TOTALS over all of its dispatches (time, HBM bytes -> aggregate rate, MFMA
busy cycles over active cycles), not medians.  Own kernels only; usage: summarize_pmc_step.py gpurun_out/pmc3"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc3"
own = ("gemm_bias_act", "act_bwd_bias", "act_fwd", "bn_", "bn2_", "sgd_momentum", "adam_apply", "embedding_", "emb_", "bucket_pack",
       "sumsq", "cast_scale", "colsum", "gemm256p", "gemm_tn", "tn_reduce", "ce_", "plan_", "maxpool", "avgpool", "attn_",
       "ln_", "filter_flip", "conv_tab", "conv3x3_c64", "conv1x1_n64", "conv1x1_stats", "wrw1x1", "stem_", "subsample",
       "conv_wrw_direct", "wd_reduce", "route_", "DtfFoldInvalid")


def short(n):
    n = n.replace("void ", "")
    if "(" in n:
        n = n[:n.index("(")]
    return n[:58]


tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
dur = collections.defaultdict(float)
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(o in r["Kernel_Name"] for o in own):
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for f in sorted(glob.glob(os.path.join(d, "mfma_kernel_trace.csv"))):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(o in r["Kernel_Name"] for o in own):
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            calls[k] += 1
print("%-58s %6s %10s %7s %9s %10s %7s %7s" % ("kernel (totals over all dispatches, profiled clocks)", "calls", "time_us", "MFMA%",
                                               "LDSconf%", "HBM_MB", "TB/s", "L2hit%"))
nan = float("nan")
for k in sorted(tot, key=lambda k: -dur[k]):
    c = tot[k]
    t = dur[k]
    gui, busy = c.get("GRBM_GUI_ACTIVE", nan), c.get("SQ_VALU_MFMA_BUSY_CYCLES", nan)
    # GRBM_GUI_ACTIVE appears in two passes (mfma, fetch): halve it
    gui = gui / 2
    mf = 100.0 * busy / (gui / 8 * 1024) if gui == gui and gui > 0 and busy == busy else nan
    lc, la = c.get("SQ_LDS_BANK_CONFLICT", nan), c.get("SQ_LDS_IDX_ACTIVE", nan)
    lds = 100.0 * lc / la if la == la and la > 0 else nan
    fk, wk = c.get("FETCH_SIZE", nan), c.get("WRITE_SIZE", nan)
    mb = (2 * fk + wk) / 1024 if fk == fk and wk == wk else nan
    hit, miss = c.get("TCC_HIT_sum", nan), c.get("TCC_MISS_sum", nan)
    l2 = 100.0 * hit / (hit + miss) if hit == hit and hit + miss > 0 else nan
    print("%-58s %6d %10.1f %7.1f %9.2f %10.1f %7.2f %7.1f" % (k, calls[k], t, mf, lds, mb, mb / t if t > 0 else nan, l2))
print("\nnotes: HBM_MB = (2 x FETCH_SIZE + WRITE_SIZE) KB / 1024 (FETCH_SIZE counts 128-B requests as 64 B on gfx950);")
print("TB/s = HBM_MB / time; MFMA% = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); L2hit% from TCC_HIT / (HIT + MISS);")
print("LDSconf% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Under --pmc kernels run one at a time at profiling clocks: times are")
print("longer than in the step trace (profiles/r3_resnet50_step.txt); the ratios are what this table is for.")
