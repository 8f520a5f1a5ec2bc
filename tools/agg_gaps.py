#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv: per level of the recursion, start / end of the aggregator stream's kernels (tokens + in_proj finish,
attention, post chain, token-0 tail) and the gaps between them.  usage: agg_gaps.py DIR"""
import csv, glob, sys, statistics as st
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in csv.DictReader(open(f))]
rows.sort()
def kind(n):
    if 'attn_m32p' in n: return 'attn'
    if 'tlayer_ws_kernel<128, true, false' in n: return 'chain'
    if 'token0_dist' in n: return 'tail'
    if 'tlayer_ws_kernel<128, false, true, false, true' in n: return 'fin'
    return None
seq = [(s, e, kind(n), q) for s, e, n, q in rows if kind(n)]
gaps = {'fin->attn': [], 'attn->chain': [], 'chain->tail': []}
dur = {'fin': [], 'attn': [], 'chain': [], 'tail': []}
last = {}
for s, e, k, q in seq:
    dur[k].append((e - s) / 1e3)
    if k == 'attn' and 'fin' in last: gaps['fin->attn'].append((s - last['fin']) / 1e3)
    if k == 'chain' and 'attn' in last: gaps['attn->chain'].append((s - last['attn']) / 1e3)
    if k == 'tail' and 'chain' in last: gaps['chain->tail'].append((s - last['chain']) / 1e3)
    last[k] = e
for k, v in dur.items():
    if v: print(f"{k:6s} n={len(v):5d} median {st.median(v):7.1f} us  min {min(v):7.1f}")
for k, v in gaps.items():
    v = [x for x in v if -50 < x < 200]
    if v: print(f"gap {k:12s} n={len(v):5d} median {st.median(v):7.2f} us  min {min(v):7.2f}  p90 {sorted(v)[int(len(v)*0.9)]:7.2f}")
