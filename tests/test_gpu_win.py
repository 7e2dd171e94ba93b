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
    # the one-launch form (the kernel merges: no scan_many_kernel / merge_kernel behind it) serves every run that gives no item up
    # and has no more items than resident waves; a run that gave one up was repeated with the two kernels behind the scan
    if expect_route == 3 and failed:
        assert b.debug_win_launches() == 3, "an item was given up, but the batch was not re-run with scan_many_kernel"
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


@pytest.mark.parametrize("nterms,k", [(3, 100), (4, 200), (2, 256), (6, 100), (7, 200), (8, 256)])  # (six to eight terms with k > 64: round 6)
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


def test_one_launch_merge_and_its_fallback(tuning):
    """Round 6: scan_win_kernel merges a query's lists itself (the wave that finishes the query's last item) -- one launch per batch.
    C3's shape: one launch, no item given up, records equal the three-launch form's byte for byte.  Thick lists (win_force): windows
    with more second arrivals than the list holds give their item up, the query's count comes back NONE32 on the device and fetch
    re-runs the batch with scan_many_kernel and merge_kernel -- the caller sees the oracle's records either way."""
    seg, gix, oix = synth_pair(700_000, 33_000, seed=7)
    terms, off = bench_queries(seg, 33_000, 256, 5, seed=3)
    tuning(fused=0)
    b, hits, nh = run_batch(gix, terms, off, 10)
    assert b.debug_win_launches() == 1 and b.debug_counts()[1] == 0
    tuning(fused=0, win_fuse=0)
    b3, hits3, nh3 = run_batch(gix, terms, off, 10)
    assert b3.debug_win_launches() == 3
    assert hits.tobytes() == hits3.tobytes() and np.array_equal(nh, nh3)
    ob, onb, _ = oix.search_batch(terms, off, 10, mode="brute", threads=8)
    assert np.array_equal(nh, onb)
    for q in range(len(off) - 1):
        assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"q{q} vs brute")
    # the fallback
    c = make_corpus(200_000, 3000, seed=2, length="lognormal", mean_len=60)
    seg2 = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
    gix2, oix2 = vb.GpuIndex(seg2), orc.OracleIndex.from_arrays(seg2.meta(), seg2.arrays())
    terms2, off2 = make_queries(c, 48, 4, seed=9)
    tuning(win_force=1, fused=0, win_fuse=1)
    b = vb.Batch(gix2, 48, len(terms2), 10)
    b.set_queries(terms2, off2)
    assert b.debug_route() == 3
    b.run()
    assert b.debug_win_launches() == 1
    hits2, nh2 = b.fetch()
    items, failed = b.debug_counts()
    assert failed > 0 and b.debug_win_launches() == 3, "this shape is expected to give items up (more second arrivals than the list holds)"
    ob, onb, _ = oix2.search_batch(terms2, off2, 10, mode="brute", threads=8)
    assert np.array_equal(nh2, onb)
    for q in range(48):
        assert_bit_exact(ob[q, :onb[q]], hits2[q, :nh2[q]], what=f"fallback q{q}")
    b.run()  # (the query set stays on the three-launch form)
    assert b.debug_win_launches() == 3
    h3, n3 = b.fetch()
    assert h3.tobytes() == hits2.tobytes()


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


def test_runs_just_beyond_a_load_per_lane_in_a_batch_of_mixed_lengths(tuning):
    """Runs of about 260 postings per window -- a few beyond the 256 one load per lane stages, so that second arrivals are looked up in
    memory behind the staged part -- in a batch whose queries have two to five terms: the lanes of a shorter query's NULL terms take
    part in that look-up's unconditional loads and must stay inside the plane (they did not for one commit of round 6: a memory fault on
    C3's index with tools/mixed_batch.py, which no test of this file reached)."""
    c = make_corpus(400_000, 15_000, seed=4, length="fixed", mean_len=60)
    seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
    gix, oix = vb.GpuIndex(seg), orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    rows = []
    for nterms in (5, 2, 3, 4, 2, 5, 3):
        t, o = make_queries(c, 40, nterms, seed=30 + nterms + len(rows))
        rows += [t[o[q]:o[q + 1]] for q in range(40)]
    rng = np.random.default_rng(1)
    rows = [rows[i] for i in rng.permutation(len(rows))]
    terms = np.concatenate(rows).astype(np.uint32)
    off = np.r_[0, np.cumsum([len(r) for r in rows])].astype(np.uint32)
    tuning(win_force=1, fused=0)
    for k in (10, 100):
        check(gix, oix, terms, off, k, expect_failed=None)
    tuning(id16_plane=0, rel16_plane=0)
    gix2 = vb.GpuIndex(seg)  # ... and behind decode_id16_kernel
    tuning(id16_plane=1, rel16_plane=1)
    check(gix2, oix, terms, off, 10, expect_failed=None)


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


def test_index_without_post_id16_takes_the_window_kernel_behind_the_decode(tuning):
    """tuning id16_plane = 0 (and rel16_plane = 0: the blob, the tf / fieldnorm words and the window TABLES only): the batch's terms are
    unpacked from the reference's bit-packed blocks into the batch's scratch plane by decode_id16_kernel ahead of every
    scan_win_kernel launch (csrc/decode_id16.h).  Same route, no item given up, the records of the index with every plane -- also with
    unknown tokens, queries of fewer terms, k = 100, a second query set on the same batch object (the scratch plane grows), and
    through the pipelined boundary."""
    seg = vb.Segment.synth(400_000, 33_000, mean_len=100, len_mode=1, seed=9)
    gix_all = vb.GpuIndex(seg)
    tuning(id16_plane=0, rel16_plane=0)
    gix = vb.GpuIndex(seg)
    tuning(id16_plane=1, rel16_plane=1, fused=0)  # (fused = 0: a handful of queries takes the batched route too)
    assert gix.device_bytes <= gix_all.device_bytes - 2 * 256 * seg.meta()["n_blocks"]  # (two planes of 256 bytes per block less)
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    for nq, nterms, k in ((600, 5, 10), (90, 3, 100), (1, 2, 10)):
        terms, off = bench_queries(seg, 33_000, nq, nterms, seed=nq)
        if nq == 90:  # unknown tokens in a few queries, anywhere
            terms = terms.copy()
            terms[off[3]] = 0xfffffff0 - 1
            terms[off[8] + 2] = 0xfffffff0
            for q in (3, 8):
                terms[off[q]:off[q + 1]] = np.sort(terms[off[q]:off[q + 1]])
        hits, nh = check(gix, oix, terms, off, k)
        _, h_all, n_all = run_batch(gix_all, terms, off, k)
        assert hits.tobytes() == h_all.tobytes() and np.array_equal(nh, n_all)
    # one batch object, query sets of growing size; then the ring of the pipelined boundary
    b = vb.Batch(gix, 512, 4096, 10)
    for nq in (3, 40, 512, 7):
        terms, off = bench_queries(seg, 33_000, nq, 5, seed=100 + nq)
        b.set_queries(terms, off)
        assert b.debug_route() == 3
        b.run()
        hits, nh = b.fetch()
        want, nw = vb.search_batch(gix_all, terms, off, 10)
        assert hits.tobytes() == want.tobytes() and np.array_equal(nh, nw)
    st = vb.Stream(gix, 3, 512, 4096, 10)
    sets = [bench_queries(seg, 33_000, nq, 5, seed=200 + nq) for nq in (300, 5, 512, 64, 1)]
    got = []
    for t, o in sets:
        if st.in_flight == 3:
            got.append(st.collect())
        st.submit(t, o)
    while st.in_flight:
        got.append(st.collect())
    for (h, n), (t, o) in zip(got, sets):
        want, nw = vb.search_batch(gix_all, t, o, 10)
        assert h.tobytes() == want.tobytes() and np.array_equal(n, nw)


@pytest.mark.parametrize("force", [1, 0])
def test_random_shapes_through_the_window_kernel(tuning, force):
    """The differential rule of the reference's fuzz test (tests/fuzz:217-303: random corpus, random queries, compare with an exact
    scorer) on the window kernel: sixteen random shapes -- 3 k .. 400 k documents, vocabularies from 50 to 40 k tokens (runs from
    empty to thousands of postings per window), uniform and Zipf tokens, one to eight terms with unknown tokens among them, one to
    ninety queries, k from 1 to 256 -- with win_force so that every sparse batch takes it whatever its density (force = 1), and with the library's own routing (force = 0).  Items the kernel
    gives up go to scan_many_kernel; the records are the oracle's either way."""
    # (VBM25_FUZZ_SEED / VBM25_FUZZ_CASES: other shapes than the suite's fixed sixteen -- tools/fuzz_more.sh runs a few hundred on the GPU box)
    rng = np.random.default_rng(int(os.environ.get("VBM25_FUZZ_SEED", 20260927)) + force)
    if force:
        tuning(win_force=1, fused=0, dense_x1000=10 ** 9)
    else:  # (the library's own routing: window, range, dense or many-term kernels as the shapes fall)
        tuning(fused=0)
    for case in range(int(os.environ.get("VBM25_FUZZ_CASES", 16))):
        n_docs = int(rng.choice([3000, 20_000, 66_000, 150_000, 400_000]))
        vocab = int(rng.choice([50, 400, 3000, 40_000]))
        if n_docs * 30 // vocab > 600_000:  # (keep the oracle's brute force in seconds)
            vocab *= 10
        zipf = None if rng.random() < 0.6 else 1.0
        c = make_corpus(n_docs, vocab, seed=100 + case, length=str(rng.choice(["fixed", "lognormal", "mixed"])), mean_len=int(rng.choice([12, 30, 60])), zipf=zipf)
        seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
        gix, oix = vb.GpuIndex(seg), orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
        nterms = int(rng.integers(1, 9))
        nq = int(rng.integers(1, 91))
        k = int(rng.choice([1, 3, 10, 64, 65, 128, 256]))
        terms, off = make_queries(c, nq, nterms, seed=case, zipf=zipf)
        if rng.random() < 0.5 and nterms > 1:  # queries of different lengths in one batch: every one keeps a random prefix of its terms
            rows = [terms[off[q]:off[q + 1]][:int(rng.integers(1, nterms + 1))] for q in range(nq)]
            terms = np.concatenate(rows).astype(np.uint32)
            off = np.r_[0, np.cumsum([len(r) for r in rows])].astype(np.uint32)
        b = vb.Batch(gix, nq, max(1, len(terms)), k)
        b.set_queries(terms, off)
        route = b.debug_route()
        b.run()
        hits, nh = b.fetch()
        ob, onb, _ = oix.search_batch(terms, off, k, mode="brute", threads=8)
        assert np.array_equal(nh, onb), (case, route, n_docs, vocab, nterms, nq, k)
        for q in range(nq):
            assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"case {case} (route {route}, {n_docs} docs, vocab {vocab}, {nterms} terms, k {k}) q{q}")
        assert route in (3, 0, 2), route
        # ... and through an index of the same segment without the post_id16 and post_rel16 planes (the scratch plane of decode_id16_kernel
        # in front of the window kernel, scan_range_kernel's in-kernel decode): the same records
        tuning(id16_plane=0, rel16_plane=0)
        gix2 = vb.GpuIndex(seg)
        tuning(id16_plane=1, rel16_plane=1)
        b2 = vb.Batch(gix2, nq, max(1, len(terms)), k)
        b2.set_queries(terms, off)
        b2.run()
        h2, n2 = b2.fetch()
        assert np.array_equal(n2, nh), (case, route, b2.debug_route(), n_docs, vocab, nterms, nq, k)
        for q in range(nq):  # (the rows behind a query's count are not part of the result)
            assert h2[q, :nh[q]].tobytes() == hits[q, :nh[q]].tobytes(), (case, route, b2.debug_route(), n_docs, vocab, nterms, nq, k, q)


@pytest.mark.parametrize("nq", [1, 2, 3, 5, 7, 13, 22])
def test_three_parts_per_query_with_any_number_of_queries(tuning, nq):
    """An index of exactly three windows: three parts per query, the layout of the skewed parts (a workgroup takes four queries, part
    k goes to the wave slots 4 k .. 4 k + 3).  Query counts that are not multiples of four: the left-over queries' items follow in
    plain order (the fuzz test above found them written beyond the end of the host's order array)."""
    seg, gix, oix = synth_pair(150_000, 12_000, mean_len=40, seed=3)
    terms, off = bench_queries(seg, 12_000, nq, 3, seed=nq)
    tuning(fused=0)
    check(gix, oix, terms, off, 10)


@pytest.mark.parametrize("zipf_s", [0, 1])
def test_unusual_batches_on_the_full_size_corpus(zipf_s):
    """tools/stress_shapes.py: on the 10 M-document index of C3 (zipf_s = 0) and of its Zipf variant C3z (zipf_s = 1) -- batches of mixed
    query lengths (1 .. 8 terms) at k = 10 / 64 / 100 / 256, through the index without post_id16 / post_rel16 too, unknown tokens and an
    empty query, batches of 1 / 7 / 1023 / 3000 queries, k = 1000, nine to twelve terms, the pipelined ring -- a sample of every batch
    against the oracle's brute force bit for bit, every batch run twice.  (A process of its own: a GPU fault aborts the process, and the
    tool announces every step before it runs.)"""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_shapes.py"), str(zipf_s)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "all shapes ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
