#!/usr/bin/env python3
"""Print the instruction-class schedule of the hottest basic block (most MFMAs) of a kernel in a .s file.
usage: asm_loop.py file.s kernel_name_substring"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in s[start + 1:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); cur, name = [], l
    else:
        cur.append(l)
blocks.append((name, cur))
best = max(blocks, key=lambda b: sum('v_mfma' in l for l in b[1]))
print(best[0], 'instr', len([l for l in best[1] if l.strip() and not l.strip().startswith(';')]), 'mfma', sum('v_mfma' in l for l in best[1]))
seq = []
for l in best[1]:
    t = l.strip().split(' ')[0]
    if t.startswith(('v_mfma', 'ds_read', 'ds_write', 's_waitcnt', 'global_load', 's_barrier', 'buffer_load', 's_cbranch')):
        extra = l.strip()[len('s_waitcnt'):] if t == 's_waitcnt' else ''
        seq.append(re.sub(r'v_mfma_\w+', 'MFMA', t) + extra)
out, prev, cnt = [], None, 0
for t in seq + [None]:
    if t == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{prev}x{cnt}" if cnt > 1 else prev)
        prev, cnt = t, 1
print(' | '.join(out))
