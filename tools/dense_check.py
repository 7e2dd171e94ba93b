#!/usr/bin/env python3
"""Zipf workload (C5 shape at a chosen size) through scan_dense_kernel: step time of the variants (MaxScore
split on / off, exhaustive scan_many_kernel), items that fell back, parity of a sample against the oracle's
brute force with the first difference spelled out.
usage: dense_check.py <n_docs> <vocab> <nq> <nterms> <k> [cache]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vectorchord_bm25_amd as vb
from vectorchord_bm25_amd import _lib

if os.environ.get("VBM25_SO"):  # a variant build of the library (tuning experiments)
    _lib._SO = os.path.join(ROOT, "vectorchord-bm25_amd", "csrc", os.environ["VBM25_SO"])
    _lib._lib = None
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


def run(env, steps=3):
    for kk, v in env.items():
        os.environ[kk] = v
    b = vb.Batch(gix, nq, len(terms), k)
    b.set_queries(terms, off)
    b.run()
    hits, nh = b.fetch()
    b.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    hits, nh = b.fetch()
    dt = (time.perf_counter() - t0) / steps
    kms, _ = b.kernel_ms()
    ni, nf = b.debug_counts()
    for kk in env:
        del os.environ[kk]
    print(f"{env}: {1e3 * dt:.2f} ms / batch of {nq} = {nq / dt:.0f} q/s (scan kernels {kms:.2f} ms); algorithmic "
          f"{algo / dt / 1e9:.0f} GB/s; items {ni}, handed to scan_many {nf}", flush=True)
    return hits, nh


variants = [{}, {"VBM25_NE": "0"}]
for extra in os.environ.get("DENSE_VARIANTS", "").split(";"):
    if extra:
        variants.append(dict(kv.split("=") for kv in extra.split(",")))
if os.environ.get("DENSE_NO_OLD") != "1":
    variants.append({"VBM25_DENSE": "0"})
results = [run(v, steps=int(os.environ.get("DENSE_STEPS", "3"))) for v in variants]
hits, nh = results[0]
for v, (h, n) in zip(variants[1:], results[1:]):
    same = h.tobytes() == hits.tobytes() and np.array_equal(n, nh)
    print(f"{v}: {'same records as the default' if same else 'RECORDS DIFFER from the default'}", flush=True)
import orc
oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
ns = min(nq, int(os.environ.get("DENSE_SAMPLE", "32")))
t0 = time.perf_counter()
ob, onb, _ = oix.search_batch(terms[:off[ns]], off[:ns + 1], k, mode="brute", threads=usable_cpus())
print(f"oracle brute force: {ns} queries in {time.perf_counter() - t0:.1f} s", flush=True)
for name, (h, n) in zip(["default"] + [str(v) for v in variants[1:]], results):
    bad = 0
    for q in range(ns):
        g, r = h[q, :n[q]], ob[q, :onb[q]]
        if n[q] == onb[q] and np.array_equal(g["doc_id"], r["doc_id"]) and np.array_equal(g["score"].view(np.uint64), r["score"].view(np.uint64)):
            continue
        bad += 1
        if bad <= 3:
            m = min(len(g), len(r))
            d = np.flatnonzero((g["doc_id"][:m] != r["doc_id"][:m]) | (g["score"][:m].view(np.uint64) != r["score"][:m].view(np.uint64)))
            i = int(d[0]) if len(d) else m
            print(f"  {name} q{q}: {n[q]} vs {onb[q]} hits; first difference at rank {i}: got "
                  f"{[(int(x['doc_id']), float(x['score'])) for x in g[i:i + 3]]} expected "
                  f"{[(int(x['doc_id']), float(x['score'])) for x in r[i:i + 3]]}; missing docs "
                  f"{sorted(set(int(x) for x in r['doc_id']) - set(int(x) for x in g['doc_id']))[:8]} extra "
                  f"{sorted(set(int(x) for x in g['doc_id']) - set(int(x) for x in r['doc_id']))[:8]}", flush=True)
    print(f"parity vs brute force, {name}: {ns - bad}/{ns} queries bit-exact", flush=True)
