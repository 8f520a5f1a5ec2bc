#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)."""
import csv, glob, collections, re, sys
f = glob.glob(sys.argv[1] + '/*/*_counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
KEYS = ('EpiLstmO','EpiLstmC','EpiLstmH','EpiImpProj','attn_f32','tlayer','token0_tail','topk','expand','gather_kernel')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = next((k for k in KEYS if k in r['Kernel_Name']), None)
    if not name: continue
    if name == 'tlayer': name += '_g' + r['Grid_Size']
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[name]['_dur'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, f"dur={m['_dur']:.0f}us", ' '.join(f"{c}={v:.4g}" for c, v in m.items() if c != '_dur'))
