#!/usr/bin/env python3
"""Zipf workload (C5 shape at a chosen size): parity of a sample against the oracle, step time with and
without the MaxScore split, work items that fell back to scan_many_kernel, CPU Block-WAND rate.
usage: zipf_check.py <n_docs> <vocab> <nq> <nterms> <k> [cache]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vectorchord_bm25_amd as vb
from bench import make_queries, usable_cpus

n_docs, vocab, nq, nterms, k = (int(x) for x in sys.argv[1:6])
cache = sys.argv[6] if len(sys.argv) > 6 else ""
t0 = time.perf_counter()
if cache and os.path.exists(cache):
    seg = vb.Segment.load(cache)
else:
    seg = vb.Segment.synth(n_docs, vocab, mean_len=100, len_mode=1, zipf_s=1.0, seed=20260925, threads=usable_cpus())
    if cache:
        seg.save(cache)
print(f"segment: {time.perf_counter() - t0:.1f} s, postings {int(seg.arrays()['term_df'].astype(np.int64).sum())}, blocks {seg.n_blocks}", flush=True)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=1.0)
algo = sum(seg.query_bytes(terms[off[q]:off[q + 1]], k) for q in range(nq))
L = vb.lib()
L.vbm25_batch_debug_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]


def run(env, steps=3):
    for kk, v in env.items():
        os.environ[kk] = v
    b = vb.Batch(gix, nq, len(terms), k)
    b.set_queries(terms, off)
    b.run()
    hits, nh = b.fetch()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    hits, nh = b.fetch()
    dt = (time.perf_counter() - t0) / steps
    ni, nf = C.c_uint32(), C.c_uint32()
    L.vbm25_batch_debug_counts(b.h, C.byref(ni), C.byref(nf))
    for kk in env:
        del os.environ[kk]
    print(f"{env}: {1e3 * dt:.2f} ms / batch of {nq} = {nq / dt:.0f} q/s; algorithmic {algo / dt / 1e9:.0f} GB/s; items {ni.value}, handed to scan_many {nf.value}", flush=True)
    return hits, nh


hits, nh = run({})
if os.environ.get("ZIPF_NO_NE") != "1":
    h2, n2 = run({"VBM25_RANGE_DENSE": "1"}, steps=1)
    assert h2.tobytes() == hits.tobytes() and np.array_equal(nh, n2), "MaxScore split changed the results"
import orc
oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
ns = min(nq, int(os.environ.get("ZIPF_SAMPLE", "32")))
t0 = time.perf_counter()
ob, onb, _ = oix.search_batch(terms[:off[ns]], off[:ns + 1], k, mode="brute", threads=usable_cpus())
print(f"oracle brute force: {ns} queries in {time.perf_counter() - t0:.1f} s", flush=True)
bad = 0
for q in range(ns):
    if not (nh[q] == onb[q] and np.array_equal(hits[q, :nh[q]]["doc_id"], ob[q, :onb[q]]["doc_id"]) and
            np.array_equal(hits[q, :nh[q]]["score"].view(np.uint64), ob[q, :onb[q]]["score"].view(np.uint64))):
        bad += 1
print(f"parity vs brute force: {ns - bad}/{ns} queries bit-exact", flush=True)
t0 = time.perf_counter()
_, _, dt = oix.search_batch(terms[:off[ns]], off[:ns + 1], k, mode="wand", threads=usable_cpus())
print(f"CPU Block-WAND restatement: {ns / dt:.1f} q/s on {usable_cpus()} threads", flush=True)
