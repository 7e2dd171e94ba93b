"""scan_win_kernel (the document-window formulation of the sparse scan: the dominant kernel of C3) against the oracle, through the
C ABI.  -m gpu only.  Bit-exact rule as in test_gpu_search.py: doc ids, score bits and payloads equal the canonical brute force."""
import os
import sys

import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus, make_queries
from parity import assert_bit_exact, assert_same_ranking

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_batch(gix, terms, off, k, expect_route=3):
    b = vb.Batch(gix, len(off) - 1, max(1, len(terms)), k)
    b.set_queries(terms, off)
    assert b.debug_route() == expect_route, f"route {b.debug_route()} instead of {expect_route}"
    b.run()
    hits, nh = b.fetch()
    return b, hits, nh


def check(gix, oix, terms, off, k, expect_route=3, wand=False, expect_failed=0):
    b, hits, nh = run_batch(gix, terms, off, k, expect_route)
    items, failed = b.debug_counts()
    if expect_failed is not None:
        assert failed == expect_failed, f"{failed} of {items} items went to scan_many_kernel"
    ob, onb, _ = oix.search_batch(terms, off, k, mode="brute", threads=8)
    assert np.array_equal(nh, onb)
    for q in range(len(off) - 1):
        assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"q{q} vs brute")
    if wand:
        ow, onw, _ = oix.search_batch(terms, off, k, mode="wand", threads=8)
        for q in range(len(off) - 1):
            t = terms[off[q]:off[q + 1]]
            assert_same_ranking(ow[q, :onw[q]], hits[q, :nh[q]], ref_ext=oix.search_brute(t, k + 300), what=f"q{q} vs wand")
    # the same batch again on the same object (the per-launch state is left clean), and through the range kernel
    b.run()
    h2, n2 = b.fetch()
    assert h2.tobytes() == hits.tobytes() and np.array_equal(n2, nh)
    return hits, nh


def synth_pair(n_docs, vocab, mean_len=100, seed=7, zipf_s=0.0):
    seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=1, zipf_s=zipf_s, seed=seed)
    return seg, vb.GpuIndex(seg), orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())


def bench_queries(seg, vocab, nq, nterms, seed, zipf_s=0.0):
    sys.path.insert(0, ROOT)
    from bench import make_queries as mq
    return mq(seg, vocab, nq, nterms, seed=seed, zipf_s=zipf_s)


@pytest.mark.parametrize("k,items", [(1, 0), (10, 0), (64, 0), (100, 0), (256, 0), (10, 192), (10, 400), (200, 192)])
def test_c3_shape_scaled_down_takes_the_window_kernel(tuning, k, items):
    """(items: work items of the batch -- 192 = one per query: all of its eleven windows in one item, the window loop's steady state)
    C3's shape (33 k vocabulary, 100 draws per document: about 190 postings per term and 2^16-document window) at 700 k documents:
    routed to scan_win_kernel by itself, no item given up, records equal the oracle's; also against the faithful Block-WAND."""
    seg, gix, oix = synth_pair(700_000, 33_000)
    terms, off = bench_queries(seg, 33_000, 192, 5, seed=3)
    if items:
        tuning(win_items=items)
    check(gix, oix, terms, off, k, wand=(k == 10 and not items))


def test_same_records_as_the_range_kernel(tuning):
    seg, gix, oix = synth_pair(500_000, 33_000, seed=11)
    terms, off = bench_queries(seg, 33_000, 200, 5, seed=4)
    _, h_win, n_win = run_batch(gix, terms, off, 10, 3)
    tuning(win=0)
    _, h_rng, n_rng = run_batch(gix, terms, off, 10, 2)
    assert h_win.tobytes() == h_rng.tobytes() and np.array_equal(n_win, n_rng)


@pytest.mark.parametrize("nterms", [1, 2, 3, 4, 5, 6, 7, 8])
def test_term_counts(tuning, nterms):
    """every query of the batch has the same number of indexed terms: the instantiation compiled for exactly that many (2 .. 8; a
    single term: the two-load kernel with a dummy)"""
    seg, gix, oix = synth_pair(300_000, 33_000, seed=5)
    terms, off = bench_queries(seg, 33_000, 64, nterms, seed=nterms)
    tuning(fused=0)
    check(gix, oix, terms, off, 10)


@pytest.mark.parametrize("nterms,k", [(3, 100), (4, 200), (2, 256)])
def test_term_counts_with_more_register_rows(tuning, nterms, k):
    seg, gix, oix = synth_pair(300_000, 33_000, seed=5)
    terms, off = bench_queries(seg, 33_000, 64, nterms, seed=10 + nterms)
    tuning(fused=0)
    check(gix, oix, terms, off, k)


@pytest.mark.parametrize("k", [10, 100])
def test_mixed_term_counts_in_one_batch(tuning, k):
    """queries of one to five terms in one batch, and one whose last token the index does not know: the kernel compiled for the most
    terms, the shorter queries load dummies"""
    seg, gix, oix = synth_pair(300_000, 33_000, seed=5)
    rows = []
    for nterms in (5, 2, 3, 1, 4, 5, 3):
        t, o = bench_queries(seg, 33_000, 12, nterms, seed=20 + nterms)
        rows += [t[o[q]:o[q + 1]] for q in range(12)]
    rows.append(np.r_[rows[0][:3], [0xfffffff0]].astype(np.uint32))  # (an id beyond the vocabulary: ignored, search.rs:59-61)
    terms = np.concatenate(rows).astype(np.uint32)
    off = np.r_[0, np.cumsum([len(r) for r in rows])].astype(np.uint32)
    tuning(fused=0)
    check(gix, oix, terms, off, k)


def test_thick_runs_are_chunked_and_searched_in_memory(tuning):
    """win_force sends lists far thicker than a load per lane holds through the kernel: runs of thousands of postings per window
    are marked in chunks and the second arrivals completed by bisection in memory; windows with more than 128 second arrivals
    hand their item to scan_many_kernel -- the records stay the oracle's either way."""
    c = make_corpus(200_000, 3000, seed=2, length="lognormal", mean_len=60)
    seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
    gix, oix = vb.GpuIndex(seg), orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    terms, off = make_queries(c, 48, 4, seed=9)
    tuning(win_force=1, fused=0)
    check(gix, oix, terms, off, 10, expect_failed=None)


def test_small_corpora_tails_and_unknown_tokens(tuning):
    """1 k .. 70 k documents (one or two windows, the last one partial), byte-packed tail blocks only or mostly, query tokens the
    index does not know, k beyond the number of matching documents."""
    tuning(win_force=1, fused=0, dense_x1000=10 ** 9)
    for n_docs, vocab, k in ((1000, 300, 10), (70_000, 4000, 64), (66_000, 50_000, 5)):
        c = make_corpus(n_docs, vocab, seed=n_docs, length="mixed", mean_len=30)
        seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
        gix, oix = vb.GpuIndex(seg), orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
        terms, off = make_queries(c, 40, 6, seed=1)
        check(gix, oix, terms, off, k, expect_failed=None)


def test_index_without_window_planes_takes_the_range_kernel(tuning):
    tuning(win_planes=0)
    seg, gix, oix = synth_pair(300_000, 33_000, seed=5)
    tuning(win_planes=1)
    terms, off = bench_queries(seg, 33_000, 200, 5, seed=2)
    check(gix, oix, terms, off, 10, expect_route=2)
