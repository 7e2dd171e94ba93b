#!/usr/bin/env python3
"""Times scan_kernel of an alternative build of the library (tuning experiments)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from vectorchord_bm25_amd import _lib

lib, cache = sys.argv[1], sys.argv[2]
_lib._SO = os.path.join(ROOT, "vectorchord-bm25_amd", "csrc", lib)
_lib._lib = None
from bench import WORKLOADS, make_queries

n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS["C3"]
seg = vb.Segment.load(cache)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=zipf_s)
b = vb.Batch(gix, nq, len(terms), k)
b.set_queries(terms, off)
for _ in range(3):
    b.run()
b.fetch()
b.set_timing(True)
for _ in range(10):
    b.run()
ms, n = b.kernel_ms()
print(f"{lib}: scan_kernel {ms:.3f} ms")
