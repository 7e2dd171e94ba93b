#!/usr/bin/env python3
"""scan_range_kernel on C3 under different work-item geometries (tuning API): kernel ms per launch, mean of 40 launches over 4 batches.
usage: tools/range_sweep.py [segment cache]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from bench import WORKLOADS, make_queries

n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS["C3"]
cache = sys.argv[1] if len(sys.argv) > 1 else ""
if cache and os.path.exists(cache):
    seg = vb.Segment.load(cache)
else:
    seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=64)
gix = vb.GpuIndex(seg)
qs = [make_queries(seg, vocab, nq, nterms, seed=1 + i, zipf_s=zipf_s) for i in range(4)]
for items in (1024, 1536, 2048, 3072, 4096):
    for grid in (512,):
        vb.reset_tuning()
        vb.set_tuning("range_items", items)
        vb.set_tuning("range_grid", grid)
        bs = []
        for t, o in qs:
            b = vb.Batch(gix, nq, len(t), k)
            b.set_queries(t, o)
            b.run()
            bs.append(b)
        for b in bs:
            b.fetch()
            b.set_timing(True)
        for i in range(40):
            bs[i % 4].run()
        tot = n = 0
        for b in bs:
            b.fetch()
            ms, c = b.kernel_ms()
            tot += ms * c
            n += c
        print(f"range_items {items:5d} grid {grid}: {tot / n:.4f} ms  work items {bs[0].debug_counts()}", flush=True)
