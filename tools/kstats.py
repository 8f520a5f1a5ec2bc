#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace csv directory: python tools/kstats.py DIR [name filter]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if flt in n:
        d[(n[:110], r.get('Grid_Size', ''))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{sum(v)/len(v):9.1f} us avg  {v[len(v)//2]:9.1f} med  {v[0]:9.1f} min  x{len(v):5d}  grid {g:>8s}  {n}")
