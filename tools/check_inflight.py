#!/usr/bin/env python3
"""check_inflight.py -- scan_win_kernel issues its loop-carried loads with inline asm and waits for them by hand (scan_win.h): the
compiler takes the loaded registers for valid the moment the asm statement ends.  This script compiles scan_win.hip to ISA and
checks, instruction by instruction in layout order, that no compiler-generated instruction READS OR WRITES a register whose
hand-issued load may still be in flight (issued by an asm global_load and not yet covered by an asm s_waitcnt that leaves fewer loads
outstanding than were issued after it).  A violation means the register allocator inserted a copy / spill of an arriving
register: restructure until it does not.  It also checks that every asm load that reads an SGPR pair is padded with `s_nop 4`
(five wait states between a VALU write of an SGPR -- v_readlane, the restore of a spilled SGPR -- and a vector memory instruction
reading it: the compiler pads its own instructions, not the inside of an asm statement).  Exit status 1 on a violation.  (Layout order is not control flow: the check is
conservative at branches -- it keeps a load in flight until a wait in layout order retires it -- which is the safe direction.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vectorchord-bm25_amd", "csrc", "scan_win.hip")


def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def check(asm_text):
    """Every kernel is scanned TWICE in layout order, the second time starting from the loads the first pass left in flight: a load
    issued at the bottom of a loop is in flight at its top (the back edge is not in the layout order)."""
    kernels, cur = [], None
    for i, line in enumerate(asm_text.split("\n")):
        t = line.strip()
        m = re.match(r"^(_ZN5vbm25\w+):", t)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
            continue
        if cur is not None:
            cur[1].append((i + 1, t))
    bad = set()
    for kernel, body in kernels:
        flight, serial = {}, 0  # register -> serial number of the load that writes it
        for rnd in range(2):
            inasm = False
            prev_in_asm = ""
            for ln, t in body:
                if t.startswith(";;#ASMSTART"):
                    inasm = True
                    prev_in_asm = ""
                    continue
                if t.startswith(";;#ASMEND"):
                    inasm = False
                    continue
                if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                    continue
                if t.startswith("s_endpgm"):
                    break
                if inasm:
                    for part in t.split("\\n"):
                        part = part.strip()
                        if part.startswith("global_load") and re.search(r"\bs\[\d+:\d+\]", part) and prev_in_asm != "s_nop 4":
                            bad.add((kernel, ln, t + "   <-- reads an SGPR pair without s_nop 4 in front (VALU write -> VMEM read hazard)"))
                        prev_in_asm = part
                        if part.startswith("global_load"):
                            serial += 1
                            for r in regs(part.split()[1].rstrip(",")):
                                flight[r] = serial
                        mm = re.search(r"vmcnt\((\d+)\)", part)
                        if mm:  # at most k loads outstanding: the k newest
                            k = int(mm.group(1))
                            flight = {r: s for r, s in flight.items() if s > serial - k}
                    continue
                parts = t.split(None, 1)
                if len(parts) < 2:
                    continue
                # any operand, source or destination: a register with a load in flight is neither read (it holds the old bytes)
                # nor written (the load would land on top of the new value) by anything but the wait that retires the load
                for r in regs(parts[1]):
                    if r in flight:
                        bad.add((kernel, ln, t))
                        break
    return sorted(bad)


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "scan_win.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                               "--cuda-device-only", SRC, "-o", out] + sys.argv[1:], stderr=subprocess.DEVNULL)
        bad = check(open(out).read())
    for k, ln, t in bad[:40]:
        print(f"{k}: line {ln}: {t}   <-- reads a register whose load may be in flight")
    print(f"{len(bad)} violation(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
