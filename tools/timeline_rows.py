#!/usr/bin/env python3
"""Print every kernel of one recursion (start, end, duration, queue, grid, short name) from a rocprofv3 kernel-trace CSV.
usage: tools/timeline_rows.py DIR [segment index, default -3]"""
import csv, glob, sys
f = (glob.glob(sys.argv[1] + '/*kernel_trace.csv') + glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'level0_kernel' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
seg = rows[idx[which]:idx[which + 1]]
t0 = int(seg[0]['Start_Timestamp'])
KEYS = ('EpiLstmO', 'EpiLstmC', 'EpiLstmH', 'EpiImpProj', 'EpiBias', 'attn_x6_prep', 'attn_x6_kernel', 'attn_token0', 'tlayer_h3',
        'token0_tail', 'topk', 'expand', 'gather_kept', 'gather', 'level0', 'Fill', 'pe_table', 'pack', 'Cat', 'copy')
for r in seg:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    n = next((k for k in KEYS if k in r['Kernel_Name']), r['Kernel_Name'][:40])
    print(f"{s/1e3:8.1f} {e/1e3:8.1f} {(e-s)/1e3:7.1f}  q={r['Queue_Id']:3s} grid={r['Grid_Size_X']:>6}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} {n}")
