#!/usr/bin/env python3
"""Where a kernel's instructions, SGPR spills (v_writelane/v_readlane) and scratch traffic come from, by source line.
usage: tools/asm_lines.py <kernel name substring> [extra hipcc flags]   (compiles csrc/search.hip with -gline-tables-only -S)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vectorchord-bm25_amd", "csrc")
kern = sys.argv[1]
out = os.path.join(ROOT, "build", "scratch", "search_g.s")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-gline-tables-only",
                "-S", "--cuda-device-only", "-o", out, "search.hip"] + sys.argv[2:], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
    files[m.group(1)] = (m.group(3) or m.group(2))
for f in re.split(r'\n(?=_ZN5vbm25\w+:)', txt):
    name = f.split(':', 1)[0]
    if kern not in name:
        continue
    cur = None
    wl, rl, sc, tot = (collections.Counter() for _ in range(4))
    for l in f.split('\n'):
        m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            cur = (files.get(m.group(1), m.group(1)).split('/')[-1], int(m.group(2)))
            continue
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if not m or cur is None or l.strip().startswith(('.', ';')):
            continue
        op = m.group(1)
        tot[cur] += 1
        if op == 'v_writelane_b32': wl[cur] += 1
        if op == 'v_readlane_b32': rl[cur] += 1
        if op.startswith('scratch_'): sc[cur] += 1
    print(name, sum(tot.values()), "instructions;", sum(wl.values()), "v_writelane,", sum(rl.values()), "v_readlane,", sum(sc.values()), "scratch")
    print("v_writelane by source line:", wl.most_common(30))
    print("v_readlane by source line:", rl.most_common(40))
    print("scratch by source line:", sc.most_common(30))
    reg = collections.Counter()
    for (fn, ln), v in tot.items():
        reg[(fn, ln // 25 * 25)] += v
    print("instructions by 25-line region:")
    for k, v in sorted(reg.items()):
        print("  ", k, v)
    break
