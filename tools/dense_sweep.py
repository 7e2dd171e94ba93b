#!/usr/bin/env python3
"""scan_dense_kernel on C5 under different work-item counts (tuning API): kernel ms per launch, mean of 6 launches over 2 batches.
usage: tools/dense_sweep.py [segment cache]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from bench import WORKLOADS, make_queries, usable_cpus

n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS["C5"]
cache = sys.argv[1] if len(sys.argv) > 1 else ""
if cache and os.path.exists(cache):
    seg = vb.Segment.load(cache)
else:
    seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=usable_cpus())
    if cache:
        seg.save(cache)
gix = vb.GpuIndex(seg)
qs = [make_queries(seg, vocab, nq, nterms, seed=1 + i, zipf_s=zipf_s) for i in range(2)]
for items in (4096, 2048, 3072, 6144, 8192, 4096):
    vb.reset_tuning()
    vb.set_tuning("dense_items", items)
    bs = []
    for t, o in qs:
        b = vb.Batch(gix, nq, len(t), k)
        b.set_queries(t, o)
        b.run()
        bs.append(b)
    for b in bs:
        b.fetch()
        b.set_timing(True)
    for i in range(6):
        bs[i % 2].run()
    tot = n = 0
    for b in bs:
        b.fetch()
        ms, c = b.kernel_ms()
        tot += ms * c
        n += c
    print(f"dense_items {items:5d}: {tot / n:.2f} ms  work items {bs[0].debug_counts()}", flush=True)
