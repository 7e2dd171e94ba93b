"""scan_win_kernel issues its loop-carried loads with inline asm and waits for them by hand (vectorchord-bm25_amd/csrc/scan_win.h):
the compiler takes an asm statement's outputs for valid the moment it ends.  tools/check_inflight.py compiles the kernel to ISA
(no GPU needed) and checks that nothing the compiler generated reads or writes a register whose load may still be in flight, and
that every asm load that reads an SGPR pair is padded against the VALU-write -> VMEM-read hazard.  Both failures were met while the
kernel was written (register copies at a loop header; a fault on a stale pointer in the eight-load kernel)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_register_with_a_load_in_flight_is_touched_and_every_sgpr_read_is_padded():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("0 violation(s)"), r.stdout[-2000:] + r.stderr[-2000:]
