"""The scalar model of scan_cursor_kernel's control logic (tools/cursor_model.py) against brute force:
block order, hashed marks with wipes, two staged blocks per term, pending list, done bits, cold pass,
chunk edges.  Host logic only; the kernel itself is covered by the -m gpu parity tests."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import cursor_model


def test_cursor_model_matches_brute_force():
    for seed in range(400):
        cursor_model.trial(seed)
