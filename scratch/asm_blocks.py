#!/usr/bin/env python3
"""usage: asm_blocks.py <file.s> <kernel-name-substring>: per basic block of the kernel: MFMAs, scratch loads / stores, VALU, waits"""
import re, sys
f, pat = sys.argv[1], sys.argv[2]
lines = open(f).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pat in l]
for start in starts:
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    print(lines[start][:100], end - start, "lines")
    cur = ['entry', 0, 0, 0, 0, 0]
    out = []
    for l in lines[start:end]:
        t = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', t):
            out.append(cur); cur = [t.split(':')[0] + (' LOOP' if 'Loop' in t else ''), 0, 0, 0, 0, 0]
        elif t.startswith('v_mfma'): cur[1] += 1
        elif t.startswith('scratch_load'): cur[2] += 1
        elif t.startswith('scratch_store'): cur[3] += 1
        elif t.startswith('s_waitcnt'): cur[5] += 1
        elif t.startswith('v_'): cur[4] += 1
    out.append(cur)
    print("%-22s %6s %6s %6s %6s %6s" % ("block", "mfma", "sld", "sst", "valu", "wait"))
    for b in out:
        if b[1] > 8 or b[2] or b[3]:
            print("%-22s %6d %6d %6d %6d %6d" % tuple(b))
