#!/usr/bin/env python3
"""Summarise one recursion from a rocprofv3 kernel-trace CSV: per-kernel durations by call position + idle gaps."""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv') + glob.glob(sys.argv[1] + '/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'level0_kernel' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
seg = rows[idx[which]:idx[which + 1]]
t0 = int(seg[0]['Start_Timestamp']); prev_end = t0; tot_gap = 0; busy = 0
KEYS = ('attn_x6_prep','attn_x6','EpiLstmO','EpiLstmC','EpiLstmH','EpiImpProj','attn_f32','tlayer','token0_tail','final_head','topk','expand','gather','level0','copyBuffer','Fill','Cat')
agg = {}
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = next((k for k in KEYS if k in r['Kernel_Name']), r['Kernel_Name'][:30])
    gap = (s - prev_end) / 1e3
    if gap > 2: print(f"  gap {gap:7.1f} us before {name} at {(s-t0)/1e3:9.1f}")
    tot_gap += max(gap, 0); busy += (e - s) / 1e3
    agg.setdefault(name, []).append((e - s) / 1e3)
    prev_end = max(prev_end, e)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:12s} n={len(v):3d} total={sum(v):8.1f} us  each={' '.join(f'{x:.0f}' for x in v[:12])}")
print(f"span {(prev_end - t0)/1e3:.1f} us  busy {busy:.1f}  gaps {tot_gap:.1f}")
