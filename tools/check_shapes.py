#!/usr/bin/env python3
"""Timing / sanity of the GPU path on other shapes than the headline one (no oracle here):
C2 (1M docs, single 3-term query), k = 100 / 1000, Zipf corpus with 10-term queries,
many-term queries (scan_many_kernel)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from bench import make_queries


def run(gix, terms, off, k, reps=5):
    b = vb.Batch(gix, len(off) - 1, len(terms), k)
    b.set_queries(terms, off)
    b.run()
    b.fetch()
    b.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        b.run()
    hits, nh = b.fetch()
    dt = (time.perf_counter() - t0) / reps
    ms, _ = b.kernel_ms()
    s = hits["score"]
    assert (s[:, :-1] >= s[:, 1:]).all()
    return dt * 1e3, ms, nh


seg = vb.Segment.synth(1_000_000, 30000, mean_len=100, len_mode=1, threads=16)
gix = vb.GpuIndex(seg)
for nq, nt, k in [(1, 3, 10), (1024, 5, 10), (1024, 5, 100), (256, 5, 1000), (64, 100, 10)]:
    terms, off = make_queries(seg, 30000, nq, nt, seed=3, zipf_s=0.0)
    wall, ms, nh = run(gix, terms, off, k)
    print(f"uniform 1M docs: {nq:5d} x {nt:3d}-term top-{k:<4d}: {wall:8.3f} ms/batch (scan kernel {ms:.3f} ms), hits/query {nh.mean():.1f}")
del gix, seg
seg = vb.Segment.synth(2_000_000, 100000, mean_len=100, len_mode=1, zipf_s=1.0, threads=16)
gix = vb.GpuIndex(seg)
for nq, nt, k in [(256, 10, 100), (256, 10, 10), (256, 3, 10)]:
    terms, off = make_queries(seg, 100000, nq, nt, seed=3, zipf_s=1.0)
    bytes_ = sum(seg.query_bytes(terms[off[q]:off[q + 1]], k) for q in range(nq))
    wall, ms, nh = run(gix, terms, off, k, reps=3)
    print(f"zipf 2M docs: {nq:5d} x {nt:3d}-term top-{k:<4d}: {wall:8.3f} ms/batch (scan kernel {ms:.3f} ms), "
          f"{bytes_ / 1e6:.0f} MB algorithmic -> {bytes_ / (wall * 1e-3) / 1e9:.1f} GB/s")
