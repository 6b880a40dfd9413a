#!/usr/bin/env python3
"""Basic-block instruction mix of one kernel in a hipcc -S dump.  Usage: isa_blocks.py file.s <mangled-name-substring>"""
import re, sys
txt = open(sys.argv[1]).read()
m = re.search(r'\n(\w*' + re.escape(sys.argv[2]) + r'\w*):\s*;.*?\n\.Lfunc_end\d+:', txt, re.S)
body = m.group(0)
blocks = []; cur = ['entry', 0, 0, 0, [], '']
for ln in body.split('\n'):
    mm = re.match(r'^(\.LBB\d+_\d+):(.*)', ln)
    if mm:
        blocks.append(cur); cur = [mm.group(1), 0, 0, 0, [], mm.group(2).strip()]; continue
    t = ln.strip()
    if t.startswith('v_'): cur[1] += 1
    elif t.startswith('s_'):
        cur[2] += 1
        mb = re.match(r's_c?branch\S*\s+(\.LBB\d+_\d+)', t)
        if mb: cur[4].append(mb.group(1))
    elif re.match(r'(global|ds|buffer|flat|scratch)_', t): cur[3] += 1
blocks.append(cur)
idx = {b[0]: i for i, b in enumerate(blocks)}
tot = [0, 0, 0]
for i, b in enumerate(blocks):
    back = [t for t in b[4] if idx.get(t, 1 << 30) <= i]
    for k in range(3): tot[k] += b[1 + k]
    print(f"{i:3d} {b[0]:12s} valu {b[1]:4d} salu {b[2]:4d} mem {b[3]:3d} {'LOOP->' + ','.join(back) if back else '':18s} {b[5][:60]}")
print("total valu/salu/mem", tot)
