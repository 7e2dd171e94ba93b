#!/usr/bin/env python3
"""Build time of the sealed segment, host builder vs device builder, on mappings of C2's shape
(1M docs x 100 uniform draws of 30k tokens; ~96M postings).  usage: flush_timing.py [n_docs]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vectorchord_bm25_amd as vb
from corpus import make_corpus

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t0 = time.perf_counter()
c = make_corpus(n_docs, 30000, seed=1, length="lognormal", mean_len=100)
print(f"mappings: {len(c['post_doc'])} postings in {time.perf_counter() - t0:.1f} s (numpy)", flush=True)
args = (c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
vb.Segment.build_device(1.2, 0.75, *[a[:10] if i < 2 else a for i, a in enumerate(args)][:0] or args)  # warm the context
for name, fn in (("host builder (all cores)", lambda: vb.Segment.build(1.2, 0.75, *args)),
                 ("device builder (incl. PCIe both ways)", lambda: vb.Segment.build_device(1.2, 0.75, *args))):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        seg = fn()
        ts.append(time.perf_counter() - t0)
    print(f"{name}: {min(ts):.3f} s  ({len(c['post_doc']) / min(ts) / 1e6:.0f} M postings/s)", flush=True)

# the same from SHUFFLED (token, document, tf) triples: numpy's sort in front of the host builder vs the device's radix sort
ts_ = c["term_start"].astype(np.int64)
term = np.repeat(np.arange(len(ts_) - 1, dtype=np.uint32), np.diff(ts_))
perm = np.random.default_rng(3).permutation(len(term))
term, doc, tf = term[perm], c["post_doc"][perm], c["post_tf"][perm]
t0 = time.perf_counter()
order = np.argsort(term.astype(np.uint64) << np.uint64(32) | doc.astype(np.uint64), kind="stable")
sdoc, stf = doc[order], tf[order]
t_sort = time.perf_counter() - t0
print(f"unsorted triples: numpy sort of the 64-bit keys {t_sort:.3f} s (then the host or device builder above)", flush=True)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    seg = vb.Segment.build_device_unsorted(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], term, doc, tf)
    ts.append(time.perf_counter() - t0)
print(f"unsorted triples: device radix sort + encode (incl. PCIe both ways): {min(ts):.3f} s  ({len(term) / min(ts) / 1e6:.0f} M mappings/s)", flush=True)
