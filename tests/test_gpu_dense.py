"""scan_dense_kernel (dense doc windows: 16-bit fixed-point upper-bound sums select, exact f64 sums decide; MaxScore
split with block-max skipping; k <= 256) through the C ABI against the CPU oracle.  -m gpu only.

tuning(dense_x1000=0) declares every query dense, which sends the whole range of corpora of the other parity
tests -- sparse lists, tail blocks, raw width-32 blocks, unknown terms, ties -- through this kernel; the batch's
debug counts prove that no item fell back to scan_many_kernel."""
import os
import sys

import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus, make_queries
from parity import assert_bit_exact

pytestmark = pytest.mark.gpu


def both(c=None, seg=None):
    if seg is None:
        seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"],
                               c["post_doc"], c["post_tf"])
    return seg, vb.GpuIndex(seg), orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())


def run_batch(gix, terms, off, k, expect_failed=0):
    b = vb.Batch(gix, len(off) - 1, max(1, len(terms)), k)
    b.set_queries(terms, off)
    b.run()
    hits, nh = b.fetch()
    items, failed = b.debug_counts()
    if expect_failed is not None:
        assert failed == expect_failed, f"{failed} of {items} items fell back to scan_many_kernel"
    return hits, nh


def check_dense(gix, oix, terms, off, k, expect_failed=0):
    hits, nh = run_batch(gix, terms, off, k, expect_failed)
    ob, onb, _ = oix.search_batch(terms, off, k, mode="brute", threads=8)
    assert np.array_equal(nh, onb)
    for q in range(len(off) - 1):
        assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"q{q} vs brute")
    return hits, nh


@pytest.mark.parametrize("n_docs,vocab,nq,nterms,k", [(300_000, 20_000, 96, 10, 100), (1_000_000, 50_000, 64, 6, 10),
                                                      (200_000, 5_000, 32, 16, 128), (200_000, 5_000, 16, 12, 256)])
def test_zipf_corpora(tuning, n_docs, vocab, nq, nterms, k):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_queries as bench_queries
    seg = vb.Segment.synth(n_docs, vocab, mean_len=100, len_mode=1, zipf_s=1.0, seed=3)
    seg, gix, oix = both(seg=seg)
    terms, off = bench_queries(seg, vocab, nq, nterms, seed=5, zipf_s=1.0)
    hits, nh = check_dense(gix, oix, terms, off, k)
    # the same records without the MaxScore split, from the exhaustive kernel, and run twice (idempotence)
    h2, n2 = run_batch(gix, terms, off, k)
    assert h2.tobytes() == hits.tobytes()
    tuning(ne=0)
    h3, n3 = run_batch(gix, terms, off, k)
    assert h3.tobytes() == hits.tobytes() and np.array_equal(n3, nh)
    tuning(ne=1, dense=0)
    h4, n4 = run_batch(gix, terms, off, k, expect_failed=None)
    assert h4.tobytes() == hits.tobytes() and np.array_equal(n4, nh)


@pytest.mark.parametrize("length,zipf,nterms,k", [
    ("fixed", None, 3, 10), ("lognormal", None, 5, 10), ("mixed", None, 2, 1),
    ("lognormal", 1.0, 10, 100), ("lognormal", 1.0, 4, 7), ("fixed", None, 5, 128), ("fixed", None, 5, 200), ("lognormal", 1.0, 6, 256)])
def test_every_query_declared_dense(tuning, length, zipf, nterms, k):
    tuning(dense_x1000=0)
    c = make_corpus(20000, 2000, seed=7, length=length, mean_len=60, zipf=zipf)
    seg, gix, oix = both(c)
    terms, off = make_queries(c, 48, nterms, seed=9, zipf=zipf)
    check_dense(gix, oix, terms, off, k)


def test_edge_cases_dense(tuning):
    tuning(dense_x1000=0)
    c = make_corpus(3000, 300, seed=3, length="lognormal", mean_len=50)
    seg, gix, oix = both(c)
    nt = gix.n_terms
    df = seg.arrays()["term_df"]
    rare = int(np.argmin(df))
    # empty | unknown only | known + unknown | rare | pair | 20 terms (scan_many_kernel's: more than 16)
    many = np.arange(20, dtype=np.uint32) * 3
    terms = np.r_[np.array([nt + 7, 3, nt + 9, rare, 5, 6], dtype=np.uint32), many]
    off = np.array([0, 0, 1, 3, 4, 6, 26], dtype=np.uint32)
    hits, nh = check_dense(gix, oix, terms, off, 10)
    assert nh[0] == 0 and nh[1] == 0 and nh[2] == 10
    # k larger than the number of matches: every matching document comes back, sorted
    hits, nh = check_dense(gix, oix, np.array([rare], dtype=np.uint32), np.array([0, 1], dtype=np.uint32), 128)
    assert nh[0] == min(128, df[rare])
    # a corpus smaller than the first window
    c = make_corpus(150, 40, seed=5, length="lognormal", mean_len=20)
    seg, gix, oix = both(c)
    terms, off = make_queries(c, 16, 3, seed=2)
    check_dense(gix, oix, terms, off, 10)
    check_dense(gix, oix, terms, off, 128)
    check_dense(gix, oix, terms, off, 200)


def test_codec_corner_cases_and_ties_dense(tuning):
    tuning(dense_x1000=0)
    # bitwidth-32 raw block, df == 128 exactly, single posting, 4-byte tf (see test_segment_builder)
    n_docs = 3_000_000
    docs_a = np.r_[np.arange(64), 2_900_000 + np.arange(64) * 3].astype(np.uint32)
    docs_b = (np.arange(128) * 7 + 5).astype(np.uint32)
    docs_c = np.array([123456], dtype=np.uint32)
    docs_d = (np.arange(300) * 9000 + 17).astype(np.uint32)
    rng = np.random.default_rng(0)
    post_tf = np.r_[np.ones(128), rng.integers(1, 70000, 128), [1 << 30], rng.integers(1, 4, 300)].astype(np.uint32)
    keys = np.zeros((4, 16), dtype=np.uint8)
    keys[:, 0] = [ord("a"), ord("b"), ord("c"), ord("d")]
    rng = np.random.default_rng(1)
    seg = vb.Segment.build(1.2, 0.75, rng.integers(1, 3000, n_docs).astype(np.uint32),
                           np.zeros((n_docs, 3), dtype=np.uint16), keys,
                           np.array([0, 128, 256, 257, 557], dtype=np.uint64),
                           np.r_[docs_a, docs_b, docs_c, docs_d], post_tf)
    seg, gix, oix = both(seg=seg)
    terms = np.array([0, 1, 2, 3, 0, 1, 2, 3, 0, 3], dtype=np.uint32)
    off = np.array([0, 1, 2, 3, 4, 8, 10], dtype=np.uint32)
    for k in (10, 100, 128, 200, 256):  # (the many tiny items of this index are the dense kernel's hard case)
        check_dense(gix, oix, terms, off, k)
    # identical documents: every score ties, far more than k candidates per window (candidate rounds), the order
    # is by ascending id
    n = 40_000
    keys = np.zeros((3, 16), dtype=np.uint8)
    keys[:, 0] = [ord("a"), ord("b"), ord("c")]
    docs = np.arange(n, dtype=np.uint32)
    seg = vb.Segment.build(1.2, 0.75, np.full(n, 10, dtype=np.uint32), np.zeros((n, 3), dtype=np.uint16), keys,
                           np.array([0, n, 2 * n, 2 * n + n // 2], dtype=np.uint64), np.r_[docs, docs, docs[::2]],
                           np.ones(2 * n + n // 2, dtype=np.uint32))
    seg, gix, oix = both(seg=seg)
    terms = np.array([0, 1, 0, 1, 2, 2], dtype=np.uint32)
    off = np.array([0, 2, 5, 6], dtype=np.uint32)
    for k in (10, 100, 128):  # (windows with thousands of tied candidates hand the item to scan_many_kernel)
        hits, nh = check_dense(gix, oix, terms, off, k, expect_failed=None)
        assert list(hits[0, :k]["doc_id"]) == list(range(k))


@pytest.mark.parametrize("items", [1024, 4096])
def test_repetitions_are_byte_identical_dense(tuning, items):
    """The shape that lost hits in one round-2 build of the k <= 256 instantiation (the codec corner-case index cut into
    thousands of tiny items, every query declared dense): 50 runs per k, every one equal to the oracle byte for byte,
    no item handed to the exhaustive kernel.  tools/dense_stress.py is the long form (assertion build, thresholds)."""
    tuning(dense_x1000=0, dense_items=items)
    n_docs = 3_000_000
    docs_a = np.r_[np.arange(64), 2_900_000 + np.arange(64) * 3].astype(np.uint32)
    docs_b = (np.arange(128) * 7 + 5).astype(np.uint32)
    docs_c = np.array([123456], dtype=np.uint32)
    docs_d = (np.arange(300) * 9000 + 17).astype(np.uint32)
    rng = np.random.default_rng(0)
    post_tf = np.r_[np.ones(128), rng.integers(1, 70000, 128), [1 << 30], rng.integers(1, 4, 300)].astype(np.uint32)
    keys = np.zeros((4, 16), dtype=np.uint8)
    keys[:, 0] = [ord("a"), ord("b"), ord("c"), ord("d")]
    rng = np.random.default_rng(1)
    seg = vb.Segment.build(1.2, 0.75, rng.integers(1, 3000, n_docs).astype(np.uint32),
                           np.zeros((n_docs, 3), dtype=np.uint16), keys,
                           np.array([0, 128, 256, 257, 557], dtype=np.uint64),
                           np.r_[docs_a, docs_b, docs_c, docs_d], post_tf)
    seg, gix, oix = both(seg=seg)
    terms = np.array([0, 1, 2, 3, 0, 1, 2, 3, 0, 3], dtype=np.uint32)
    off = np.array([0, 1, 2, 3, 4, 8, 10], dtype=np.uint32)
    for k in (64, 128, 200, 256):  # (the three instantiations; 200 and 256 share the four-row one)
        ob, onb, _ = oix.search_batch(terms, off, k, mode="brute", threads=8)
        want = b"".join(ob[q, :onb[q]].tobytes() for q in range(6))
        for rep in range(50):
            hits, nh = run_batch(gix, terms, off, k, expect_failed=0)
            assert np.array_equal(nh, onb), f"k={k} run {rep}: {nh} != {onb}"
            got = b"".join(hits[q, :nh[q]].tobytes() for q in range(6))
            if got != want:  # (the records' padding bytes may differ between the two sources: compare the fields)
                for q in range(6):
                    assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"k={k} run {rep} q{q}")


def test_correlated_terms_dense(tuning):
    tuning(dense_x1000=0)
    n_docs = 600_000
    rng = np.random.default_rng(42)
    base = np.sort(rng.choice(n_docs, 9000, replace=False)).astype(np.uint32)
    lists = [base, base,
             np.sort(np.r_[base[::7], rng.choice(n_docs, 6000, replace=False)]).astype(np.uint32),
             np.sort(np.r_[base[::3], rng.choice(n_docs, 2000, replace=False)]).astype(np.uint32),
             np.sort(rng.choice(n_docs, 70000, replace=False)).astype(np.uint32)]
    lists = [np.unique(l) for l in lists]
    keys = np.zeros((len(lists), 16), dtype=np.uint8)
    keys[:, 0] = np.arange(len(lists)) + ord("a")
    term_start = np.r_[0, np.cumsum([len(l) for l in lists])].astype(np.uint64)
    post_doc = np.concatenate(lists)
    post_tf = rng.integers(1, 4, len(post_doc)).astype(np.uint32)
    seg = vb.Segment.build(1.2, 0.75, rng.integers(5, 400, n_docs).astype(np.uint32),
                           np.zeros((n_docs, 3), dtype=np.uint16), keys, term_start, post_doc, post_tf)
    seg, gix, oix = both(seg=seg)
    queries = [[0, 1], [0, 2], [0, 3], [0, 1, 2, 3], [0, 1, 2, 3, 4], [2, 3, 4], [1, 4], [0, 2, 4]]
    terms = np.array([t for q in queries for t in q], dtype=np.uint32)
    off = np.r_[0, np.cumsum([len(q) for q in queries])].astype(np.uint32)
    for k in (10, 100):
        check_dense(gix, oix, terms, off, k)


def test_mixed_batch_sparse_and_dense_queries():
    """default routing: sparse queries -> scan_range_kernel, dense ones -> scan_dense_kernel, in one batch"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_queries as bench_queries
    seg = vb.Segment.synth(400_000, 30_000, mean_len=100, len_mode=1, zipf_s=1.0, seed=11)
    seg, gix, oix = both(seg=seg)
    dt, do = bench_queries(seg, 30_000, 24, 8, seed=2, zipf_s=1.0)
    df = seg.arrays()["term_df"]
    rare_ids = np.flatnonzero((df > 50) & (df < 2000))[:120].astype(np.uint32)
    st = np.sort(rare_ids.reshape(24, 5), axis=1).reshape(-1)
    terms = np.r_[dt, st].astype(np.uint32)
    off = np.r_[do, do[-1] + (np.arange(24) + 1) * 5].astype(np.uint32)
    check_dense(gix, oix, terms, off, 10)
