#!/usr/bin/env python3
"""scan_range_kernel<128> / <256> (sparse queries of 9 .. 16 terms with 64 < k <= 256: the window route takes at most eight terms) on
C3's index: 512 queries of 12 terms, k = 100 and k = 200.  Run once per build (VBM25_LIBRARY=...): two workgroups per CU with spilled
registers (-DVBM25_RWPS_BIGK=4, rounds 2-5) against one workgroup per CU without (the round-6 default)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from bench import make_queries

seg = vb.DeviceSegment.synth(10_000_000, 30_000, mean_len=100, len_mode=1, zipf_s=0.0, seed=20260925, device=0)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, 30_000, 512, 12, seed=5, zipf_s=0.0)
for k in (100, 200):
    b = vb.Batch(gix, 512, len(terms), k)
    b.set_queries(terms, off)
    for _ in range(10):
        b.run()
    h0, n0 = b.fetch()
    b.set_timing(True)
    for _ in range(30):
        b.run()
    h, n = b.fetch()
    ms, nl = b.kernel_ms()
    items, failed = b.debug_counts()
    print(f"{os.path.basename(vb.library_path()):20s} k = {k:3d}: route {b.debug_route()}  scan kernels {ms:.4f} ms per 512-query batch ({nl} launches), items {items}, given up {failed}, "
          f"hits sha {hash(h.tobytes()) & 0xffffffff:08x}", flush=True)
