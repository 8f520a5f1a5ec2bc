#!/usr/bin/env python3
"""From a one-stream rocprofv3 --kernel-trace csv of bench.py: durations of the aggregator kernels BY LEVEL of the recursion (the i-th
launch of each kernel in a step = level i).  usage: agg_by_level.py DIR"""
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
def kind(n):
    if 'attn_m32p' in n: return 'attn'
    if 'tlayer_ws_kernel<128, true, false' in n: return 'chain'
    if 'token0_dist' in n: return 'tail'
    if 'tlayer_ws_kernel<128, false, true, false, true' in n: return 'fin'
    return None
seq = {k: [] for k in ('fin', 'attn', 'chain', 'tail')}
for s, e, n in rows:
    k = kind(n)
    if k: seq[k].append((e - s) / 1e3)
for k, v in seq.items():
    per = 4 if k == 'fin' else 5          # (the last level has no fused finish in mode 2? counts decide)
    n = len(v)
    for per in (5, 4):
        if n % per == 0: break
    lv = [[v[i] for i in range(j, n, per)][per:] for j in range(per)]      # skip the warm-up step
    print(f"{k:6s} launches {n} = {per} per step: " + "  ".join(f"L{j}: {st.median(x):6.1f}" for j, x in enumerate(lv)) + f"   mean of medians {sum(st.median(x) for x in lv) / per:6.1f}")
