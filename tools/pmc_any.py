#!/usr/bin/env python3
"""Mean of every counter per (kernel, grid) of a rocprofv3 --pmc counter_collection.csv.  usage: pmc_any.py DIR [name-substring]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
key = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if key not in r['Kernel_Name']: continue
    n = r['Kernel_Name'][:48] + '|' + r['Grid_Size']
    agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[n]['_dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, ' '.join(f"{c}={v:.4g}" for c, v in sorted(m.items())))
