#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv of a ONE-STREAM run: every kernel's duration and the gap in front of it (Start - End of the
previous dispatch), grouped by (previous kernel, kernel).  usage: seq_gaps.py DIR [min_count]"""
import csv, glob, re, sys, statistics as st
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'paths_epi::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:70]
acc = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    acc.setdefault((short(n0), short(n1)), []).append(((s1 - e0) / 1e3, (e1 - s1) / 1e3, (e1 - e0) / 1e3))
tot = 0.0
out = []
for (a, b), v in acc.items():
    if len(v) < minc: continue
    gap = st.median(x[0] for x in v); dur = st.median(x[1] for x in v); adv = st.median(x[2] for x in v)
    out.append((len(v) * adv, len(v), gap, dur, adv, a, b))
out.sort(reverse=True)
print(f"{'n':>6s} {'gap':>7s} {'dur':>8s} {'end-end':>8s}  kernel   <- previous")
for w, n, gap, dur, adv, a, b in out:
    print(f"{n:6d} {gap:7.2f} {dur:8.2f} {adv:8.2f}  {b}   <- {a}")
