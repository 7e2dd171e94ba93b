#!/usr/bin/env python3
"""Compact table of `hipcc -Rpass-analysis=kernel-resource-usage` for search.hip (or the file given):
kernel, SGPRs, VGPRs, spilled SGPRs / VGPRs, scratch bytes per lane, occupancy, LDS bytes.
Usage: tools/resources.py [extra hipcc flags ...]   (run from anywhere; compiles csrc/search.hip)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vectorchord-bm25_amd", "csrc")
src = "search.hip"
extra = [a for a in sys.argv[1:]]
for a in list(extra):
    if a.endswith(".hip"):
        src = a
        extra.remove(a)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n
print(f"{'kernel':70s} {'SGPR':>5s} {'VGPR':>5s} {'sSpill':>6s} {'vSpill':>6s} {'scratch':>7s} {'occ':>3s} {'LDS':>7s}")
for r in rows:
    n = demangle(r["name"])
    n = re.sub(r"\(.*", "", n).replace("void vbm25::", "").replace("vbm25::", "")
    if "rocprim" in n or "hipcub" in n: continue
    print(f"{n[:70]:70s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs','?'):>5s} {r.get('SGPRs Spill','?'):>6s} "
          f"{r.get('VGPRs Spill','?'):>6s} {r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>3s} "
          f"{r.get('LDS Size [bytes/block]','?'):>7s}")
