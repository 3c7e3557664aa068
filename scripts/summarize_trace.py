#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-category time of ONE steady-state
training step (delimited by the K1 shard-apply launches)."""
import collections
import csv
import sys


def cat(n):
    if n.startswith('igemm_fwd'): return 'conv fwd (MIOpen igemm asm)'
    if n.startswith('igemm_bwd'): return 'conv bwd-data (MIOpen igemm asm)'
    if n.startswith('igemm_wrw'): return 'conv bwd-weight (MIOpen igemm asm)'
    if 'kernel_grouped_conv_fwd' in n: return 'conv fwd (MIOpen ck)'
    if 'kernel_grouped_conv_bwd_data' in n: return 'conv bwd-data (MIOpen ck)'
    if 'bwd_weight' in n: return 'conv bwd-weight (MIOpen ck)'
    if 'naive_conv' in n: return 'conv naive (MIOpen find, warm-up only)'
    if 'bn_reduce_kernel<0' in n: return 'K7 bn fwd stats'
    if 'bn_reduce_kernel' in n: return 'K7 bn bwd reduce'
    if 'bn_fwd_apply' in n: return 'K7 bn fwd apply'
    if 'bn_bwd_apply' in n: return 'K7 bn bwd apply'
    if 'finalize' in n: return 'K7 bn finalize'
    if 'conv_wrw_direct' in n or 'wd_reduce' in n: return 'K14 conv weight gradient (own direct kernel + slab reduce)'
    if 'gemm256p_kernel' in n and 'true>(GemmArgs' in n: return 'K13 conv fwd / dgrad (own implicit GEMM)'
    if 'gemm_bias_act' in n or 'dtf_gemm' in n or 'gemm256p_kernel' in n: return 'K6 gemm'
    if 'embedding' in n: return 'K3/K4 embedding'
    if 'SubTensorOp' in n or 'fillBuffer' in n: return 'MIOpen aux (zero/cast wrw workspace)'
    if 'CUDAFunctor_add' in n: return 'autograd grad accumulate add'
    if 'max_pool' in n: return 'maxpool'
    if 'sgd_momentum' in n: return 'K1 sgd-momentum shard apply'
    if 'adam_apply' in n: return 'K2 adam shard apply'
    if 'bucket_pack' in n: return 'K5 bucket pack'
    if n.startswith('Cijk_'): return 'hipBLASLt/Tensile gemm'
    return n[:60]


def main(path, total_steps=7):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'sgd_momentum' in r['Kernel_Name'] or 'adam_apply' in r['Kernel_Name']]
    # shard-apply launches per step (= buckets): the smallest k for which the launches between apply[-1-k] and
    # apply[-1] repeat the launches before them name by name (the traced run's step count is not needed and the
    # bench's extra probe steps do not matter); falls back to total_steps = warm-up + timed steps
    names = [r['Kernel_Name'] for r in rows]
    per_step = 0
    for k in range(1, 17):
        if len(idx) < 2 * k + 1:
            break
        a, b = idx[-1 - k], idx[-1]
        n = b - a
        if n >= 40 and a + 1 - n >= 0 and names[a + 1:b + 1] == names[a + 1 - n:a + 1]:
            per_step = k
            break
    if not per_step:
        # launches of the compute and the communication stream interleave differently from step to step: same multiset
        for k in range(1, 17):
            if len(idx) < 2 * k + 1:
                break
            a, b = idx[-1 - k], idx[-1]
            n = b - a
            if n >= 40 and a + 1 - n >= 0 and sorted(names[a + 1:b + 1]) == sorted(names[a + 1 - n:a + 1]):
                per_step = k
                break
    if not per_step:
        per_step = max(1, round(len(idx) / float(total_steps)))
    a, b = idx[-1 - per_step], idx[-1]
    step = rows[a + 1:b + 1]
    agg = collections.OrderedDict()
    for r in step:
        k = cat(r['Kernel_Name'])
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        e = agg.setdefault(k, [0, 0.0])
        e[0] += 1
        e[1] += d
    tot = sum(v[1] for v in agg.values())
    span = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
    print("one training step: span %.3f ms, sum of kernel time %.3f ms, %d kernel launches" % (span, tot / 1e3, len(step)))
    print("%10s %6s %6s  %s" % ("time_us", "share", "calls", "category"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%10.1f %5.1f%% %6d  %s" % (v[1], 100 * v[1] / tot, v[0], k))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7)
