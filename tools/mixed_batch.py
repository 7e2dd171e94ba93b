#!/usr/bin/env python3
"""scan_win_kernel on batches of mixed query lengths (C3's index): 1024 queries of 2 .. 5 terms against 1024 queries of 5 terms.
A query's missing terms read the null window table, so one kernel -- compiled for the batch's longest query -- serves both."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from bench import WORKLOADS, make_queries

n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS["C3"]
seg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, device=0)
gix = vb.GpuIndex(seg)


def run(label, rows):
    terms = np.concatenate(rows).astype(np.uint32)
    off = np.r_[0, np.cumsum([len(r) for r in rows])].astype(np.uint32)
    b = vb.Batch(gix, len(rows), len(terms), k)
    b.set_queries(terms, off)
    for _ in range(30):
        b.run()
    b.fetch()
    b.set_timing(True)
    for _ in range(100):
        b.run()
    b.fetch()
    ms, n = b.kernel_ms()
    items, failed = b.debug_counts()
    print(f"{label:44s} route {b.debug_route()}  {len(terms):5d} terms  kernel {ms:.4f} ms ({n} launches)  items {items}, given up {failed}")


def rows_of(nt, n, seed):
    t, o = make_queries(seg, vocab, n, nt, seed=seed, zipf_s=0.0)
    return [t[o[q]:o[q + 1]] for q in range(n)]


run("1024 x 5 terms", rows_of(5, 1024, 1))
run("1024 x 3 terms", rows_of(3, 1024, 2))
mixed = rows_of(2, 256, 3) + rows_of(3, 256, 4) + rows_of(4, 256, 5) + rows_of(5, 256, 6)
rng = np.random.default_rng(0)
mixed = [mixed[i] for i in rng.permutation(len(mixed))]
run("1024 x 2 .. 5 terms (256 of each, shuffled)", mixed)
