#!/usr/bin/env python3
"""Static instruction counts between the MARK_S1_* markers of scan_range_kernel<64, 8, false> (run tools/asm_lines.py first)."""
import collections, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "build", "scratch", "search_g.s")).read()
for f in re.split(r'\n(?=_ZN5vbm25\w+:)', txt):
    name = f.split(':', 1)[0]
    if 'scan_range_kernelILi64ELi8ELb0' not in name:
        continue
    lines = f.split('\n')
    marks = [(i, l.strip()) for i, l in enumerate(lines) if 'MARK_' in l]
    for (a, na), (b, nb) in zip(marks, marks[1:]):
        c = collections.Counter()
        for l in lines[a:b]:
            m = re.match(r'\s+([a-z_0-9]+)', l)
            if not m or l.strip().startswith(('.', ';')):
                continue
            op = m.group(1)
            k = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else \
                'vmem' if op.startswith(('global', 'buffer', 'scratch', 'flat')) else 'other'
            c[k] += 1
        print(na, '->', nb, dict(c))
