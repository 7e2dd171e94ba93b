#!/usr/bin/env python3
"""Unusual batches on the FULL-SIZE C3 index (10 M documents) against the oracle's brute force, bit for bit: what the suite's small
corpora cannot reach (runs thicker than a load per lane in most items, every wave of the chip busy, 153 windows).  Every step is
announced with flush=True before it runs: a GPU fault aborts the process.
   gpurun -- 'python tools/stress_shapes.py [zipf_s]'      (0: C3's uniform vocabulary; 1: C3z's Zipf vocabulary)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vectorchord_bm25_amd as vb
from bench import WORKLOADS, make_queries, oracle_index
from parity import assert_bit_exact

zipf_s = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
n_docs, vocab, mean_len, len_mode, _, _, _, _ = WORKLOADS["C3"]
print("corpus", flush=True)
seg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, device=0)
gix = vb.GpuIndex(seg)
vb.set_tuning("id16_plane", 0)
vb.set_tuning("rel16_plane", 0)
gix_dec = vb.GpuIndex(seg)
vb.reset_tuning()
print("oracle index", flush=True)
oix = oracle_index(seg)
rng = np.random.default_rng(7)


def rows_of(nt, n, seed):
    t, o = make_queries(seg, vocab, n, nt, seed=seed, zipf_s=zipf_s)
    return [t[o[q]:o[q + 1]] for q in range(n)]


def pack(rows):
    terms = np.concatenate(rows).astype(np.uint32) if rows else np.zeros(0, dtype=np.uint32)
    off = np.r_[0, np.cumsum([len(r) for r in rows])].astype(np.uint32)
    return terms, off


def check(label, rows, k, index=None, n_check=96):
    index = index or gix
    terms, off = pack(rows)
    print(f"{label}: {len(rows)} queries, {len(terms)} terms, k = {k} ...", flush=True)
    b = vb.Batch(index, len(rows), max(1, len(terms)), k)
    b.set_queries(terms, off)
    route = b.debug_route()
    b.run()
    hits, nh = b.fetch()
    b.run()  # the same batch again: the per-launch state was left clean
    h2, n2 = b.fetch()
    assert np.array_equal(nh, n2) and all(hits[q, :nh[q]].tobytes() == h2[q, :nh[q]].tobytes() for q in range(len(rows))), "second run differs"
    sel = np.sort(rng.choice(len(rows), min(n_check, len(rows)), replace=False))
    st, so = pack([rows[q] for q in sel])
    ob, onb, _ = oix.search_batch(st, so, k, mode="brute", threads=16)
    for i, q in enumerate(sel):
        assert nh[q] == onb[i], (label, q, nh[q], onb[i])
        assert_bit_exact(ob[i, :onb[i]], hits[q, :nh[q]], what=f"{label} q{q}")
    print(f"   route {route}, items / given up {b.debug_counts()}: {len(sel)} queries equal the oracle", flush=True)
    return hits, nh


mixed = rows_of(2, 256, 3) + rows_of(3, 256, 4) + rows_of(4, 256, 5) + rows_of(5, 256, 6)
mixed = [mixed[i] for i in rng.permutation(len(mixed))]
h_all, n_all = check("2 .. 5 terms shuffled", mixed, 10)
h_dec, n_dec = check("the same through the index without post_id16 / post_rel16", mixed, 10, index=gix_dec)
assert np.array_equal(n_all, n_dec) and all(h_all[q, :n_all[q]].tobytes() == h_dec[q, :n_all[q]].tobytes() for q in range(len(mixed)))
check("2 .. 5 terms, k = 100", mixed[:600], 100)
check("2 .. 5 terms, k = 256", mixed[:300], 256)
wide = rows_of(8, 200, 7) + rows_of(6, 200, 8) + rows_of(1, 200, 9) + rows_of(7, 100, 10)
wide = [wide[i] for i in rng.permutation(len(wide))]
check("1 .. 8 terms", wide, 10)
check("1 .. 8 terms, k = 64, decode index", wide, 64, index=gix_dec)
unk = [r.copy() for r in mixed[:500]]
for q in range(0, 500, 7):  # unknown tokens: first, last, alone
    unk[q] = np.sort(np.r_[unk[q][:-1], [0xfffffff0 - q]]).astype(np.uint32)
unk[3] = np.array([0xfffffff1], dtype=np.uint32)
unk[4] = np.zeros(0, dtype=np.uint32)
check("unknown tokens and an empty query", unk, 10)
check("odd batch sizes: 1", mixed[:1], 10)
check("odd batch sizes: 7", mixed[:7], 10)
check("odd batch sizes: 1023", mixed[:1023], 10)
check("3000 queries (more items than resident waves)", (mixed * 3)[:3000], 10, n_check=64)
many = rows_of(5, 64, 11)
check("k = 1000 (scan_many_kernel)", many[:32], 1000, n_check=16)
check("nine to twelve terms (scan_range_kernel)", rows_of(9, 100, 12) + rows_of(12, 100, 13), 10, n_check=48)
print("the pipelined ring with these batches ...", flush=True)
st = vb.Stream(gix, 3, 1024, 8192, 10)
sets = [pack(mixed), pack(wide[:512]), pack(mixed[:333]), pack(unk), pack(mixed[:1])]
got = []
for t, o in sets:
    if st.in_flight == 3:
        got.append(st.collect())
    st.submit(t, o)
while st.in_flight:
    got.append(st.collect())
for (h, n), (t, o) in zip(got, sets):
    b = vb.Batch(gix, len(o) - 1, max(1, len(t)), 10)
    b.set_queries(t, o)
    b.run()
    hw, nw = b.fetch()
    assert np.array_equal(n, nw) and all(h[q, :n[q]].tobytes() == hw[q, :n[q]].tobytes() for q in range(len(o) - 1))
print("   the ring's records equal the single batches'", flush=True)
print("all shapes ok", flush=True)
