"""Compact view of a kernel's instruction stream in a hipcc -S file: one letter per instruction
(M mfma, r ds_read, w ds_write, G global_load, S global_store, | s_waitcnt, B s_barrier, v VALU, s other)."""
import re
import sys
from collections import Counter


def main(path, pat):
    s = open(path).read()
    for k in re.split(r'\n(?=_Z\w+:)', s):
        name = k.split(':')[0]
        if pat not in name:
            continue
        ops = [l.split()[0] for l in k.split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        print(name, len(ops))
        print(' '.join(f"{o}:{n}" for o, n in Counter(ops).most_common(30)))

        def cls(o):
            if o.startswith('v_mfma'): return 'M'
            if o.startswith('ds_read') or o.startswith('ds_load'): return 'r'
            if o.startswith('ds_write') or o.startswith('ds_store'): return 'w'
            if o.startswith(('global_load', 'buffer_load')): return 'G'
            if o.startswith(('global_store', 'buffer_store')): return 'S'
            if o.startswith('s_waitcnt'): return '|'
            if o.startswith('s_barrier'): return 'B'
            if o.startswith('v_'): return 'v'
            return 's'
        seq = ''.join(cls(o) for o in ops)
        for i in range(0, len(seq), 160):
            print(seq[i:i + 160])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
