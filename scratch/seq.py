#!/usr/bin/env python3
"""usage: seq.py <file.s> <kernel substring>: instruction-class strings of the kernel's MFMA-carrying basic blocks
(M mfma, s salu, L vmem load, S vmem store, d lds, w waitcnt, v valu, E transcendental, a accvgpr, n nop, b branch)"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pat in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
def cls(t):
    for p, c in (('v_mfma', 'M'), ('s_waitcnt', 'w'), ('s_nop', 'n'), ('s_cbranch', 'b'), ('s_branch', 'b'), ('s_', 's'), ('buffer_load', 'L'), ('global_load', 'L'),
                 ('buffer_store', 'S'), ('global_store', 'S'), ('ds_', 'd'), ('v_exp', 'E'), ('v_rcp', 'E'), ('v_rsq', 'E'), ('v_accvgpr', 'a'), ('v_', 'v')):
        if t.startswith(p): return c
    return '?'
cur = ['entry', []]; blocks = []
for l in lines[start:end]:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t): blocks.append(cur); cur = [t.split(':')[0], []]
    elif t and t[0] not in ';.': cur[1].append(t)
blocks.append(cur)
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for name, ins in blocks:
    m = sum(1 for t in ins if t.startswith('v_mfma'))
    if m >= lo:
        print(name, len(ins), 'mfma', m); print(''.join(cls(t) for t in ins))
