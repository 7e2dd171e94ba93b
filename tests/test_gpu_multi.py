"""The N-GPU split behind the C ABI (vbm25_multi_*) on the one GPU of the test box: replicas made with
hipMemcpyPeerAsync on device 0 itself, the batch sharded over them -- every record identical to the single-handle result.
Also: the pipelined host-buffer boundary (vbm25_stream_*).  -m gpu only."""
import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus, make_queries
from parity import assert_bit_exact

pytestmark = pytest.mark.gpu


def _setup(n_docs=120_000, vocab=3000, seed=11):
    c = make_corpus(n_docs, vocab, seed=seed, length="lognormal", mean_len=60)
    seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
    return c, seg


@pytest.mark.parametrize("n_rep,nq,k", [(2, 64, 10), (3, 101, 10), (2, 7, 10), (4, 3, 5), (2, 40, 300), (2, 6, 2000)])
def test_multi_replicas_on_one_device_match_single_handle(n_rep, nq, k):
    """Shard sizes that differ by one, fewer queries than replicas (empty shards), the one-launch route inside a shard
    (<= 8 queries), k beyond the register top-k, the exhaustive k > 1024 path: byte-identical records."""
    c, seg = _setup()
    terms, off = make_queries(c, nq, 4, seed=5)
    single = vb.GpuIndex(seg)
    h1, n1 = vb.search_batch(single, terms, off, k)
    multi = vb.MultiIndex(seg, [0] * n_rep)
    assert multi.n_devices == n_rep
    h2, n2 = multi.search_batch(terms, off, k)
    assert np.array_equal(n1, n2)
    assert h1.tobytes() == h2.tobytes()
    # the resident form, run twice (state left clean by the first run), and a second query set on the same object
    mb = vb.MultiBatch(multi, nq, len(terms), k)
    mb.set_queries(terms, off)
    for _ in range(2):
        mb.run()
        h3, n3 = mb.fetch()
        assert np.array_equal(n1, n3) and h1.tobytes() == h3.tobytes()
    terms2, off2 = make_queries(c, max(1, nq // 2), 3, seed=6)
    mb.set_queries(terms2, off2)
    mb.run()
    h4, n4 = mb.fetch()
    h5, n5 = vb.search_batch(single, terms2, off2, k)
    assert np.array_equal(n4, n5) and h4.tobytes() == h5.tobytes()


def test_multi_argument_errors():
    c, seg = _setup(5000, 200)
    with pytest.raises(vb.Vbm25Error) as e:
        vb.MultiIndex(seg, [])
    assert e.value.code == -1
    with pytest.raises(vb.Vbm25Error) as e:
        vb.MultiIndex(seg, [0, 99])
    assert e.value.code == -1
    multi = vb.MultiIndex(seg, [0, 0])
    with pytest.raises(vb.Vbm25Error):  # k = 0: "number of needed rows is set to 0" (default.rs:114-116)
        multi.search_batch(np.zeros(1, dtype=np.uint32), np.array([0, 1], dtype=np.uint32), 0)
    h, n = multi.search_batch(np.zeros(0, dtype=np.uint32), np.array([0], dtype=np.uint32), 10)  # no queries
    assert len(n) == 0


@pytest.mark.parametrize("nq", [9, 11, 33, 101])
def test_odd_query_counts_do_not_write_past_the_callers_counts(nq):
    """The kernels' records sit 8-byte aligned behind the counts in the pinned buffer; the copy into the caller's n_hits must be
    exactly 4 nq bytes (round-5 advisor: the aligned size was copied, four bytes too many for an odd nq).  n_hits is handed over as
    the first nq words of a longer array whose tail is a guard pattern: vbm25_search_batch (general route: more than 8 queries),
    vbm25_stream_collect and vbm25_multi_batch_fetch with odd shards on the general route (nq >= 18 over two replicas)."""
    import ctypes as C
    c, seg = _setup(150_000, 3000, seed=4)
    gix = vb.GpuIndex(seg)
    terms, off = make_queries(c, nq, 4, seed=9)
    L = vb.lib()
    GUARD = 0xA5A5A5A5

    def guarded():
        hits = np.zeros((nq, 10), dtype=vb.HIT_DTYPE)
        cnt = np.full(nq + 4, GUARD, dtype=np.uint32)
        return hits, cnt

    want_h, want_n = vb.search_batch(gix, terms, off, 10)
    hits, cnt = guarded()
    vb.api.check(L.vbm25_search_batch(gix.h, terms.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), nq, 10,
                                  hits.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)))
    assert np.all(cnt[nq:] == GUARD), "vbm25_search_batch wrote past n_hits[nq]"
    assert np.array_equal(cnt[:nq], want_n) and hits.tobytes() == want_h.tobytes()

    st = vb.Stream(gix, 2, nq, len(terms), 10)
    st.submit(terms, off)
    hits, cnt = guarded()
    got = C.c_uint32()
    vb.api.check(L.vbm25_stream_collect(st.h, hits.ctypes.data, cnt.ctypes.data, C.byref(got)))
    st._nq.pop(0)
    assert got.value == nq and np.all(cnt[nq:] == GUARD), "vbm25_stream_collect wrote past n_hits[nq]"
    assert np.array_equal(cnt[:nq], want_n) and hits.tobytes() == want_h.tobytes()

    for n_rep in (2, 3):
        multi = vb.MultiIndex(seg, [0] * n_rep)
        mb = vb.MultiBatch(multi, nq, len(terms), 10)
        mb.set_queries(terms, off)
        for _ in range(3):  # (a stale count of a neighbouring shard would show on a repeated run)
            mb.run()
            hits, cnt = guarded()
            vb.api.check(L.vbm25_multi_batch_fetch(mb.h, hits.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)))
            assert np.all(cnt[nq:] == GUARD), "vbm25_multi_batch_fetch wrote past n_hits[nq]"
            assert np.array_equal(cnt[:nq], want_n) and hits.tobytes() == want_h.tobytes()


def test_prefilter_by_over_fetch_matches_the_filtered_ranking():
    """prefilter = on (default.rs:120-128): the shim's over-fetch loop (vb.search_batch_filtered) against the oracle's full ranking
    with the same filter applied: the first k accepted hits, for filters that keep half, a tenth and one in 300 of the documents
    (the last one forces the deeper rounds and exhausts some queries)."""
    c, seg = _setup(200_000, 2000, seed=8)
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    gix = vb.GpuIndex(seg)
    terms, off = make_queries(c, 24, 3, seed=12)
    k = 10
    for modulus in (2, 10, 300):
        keep = lambda h, m=modulus: (h["doc_id"] % m) == 1  # noqa: E731
        hits, nh, rounds = vb.search_batch_filtered(gix, terms, off, k, keep)
        assert rounds >= (1 if modulus == 2 else 2)
        for q in range(24):
            t = terms[off[q]:off[q + 1]]
            full = oix.search_brute(t, 65535)
            want = full[keep(full)][:k]
            assert nh[q] == len(want)
            assert_bit_exact(want, hits[q, :nh[q]], what=f"filter 1/{modulus} q{q}")


def test_pipelined_boundary_returns_the_single_batch_records():
    """vbm25_stream_*: batches of different shapes interleaved through a ring of three slots (upload of one, scan of another and
    download of a third in flight at once) come back first in first out with the records vbm25_search_batch gives for each
    alone; submitting beyond the depth and collecting from an empty ring are argument errors."""
    seg = vb.Segment.synth(400_000, 33_000, mean_len=100, len_mode=1, seed=9)
    gix = vb.GpuIndex(seg)
    rng = np.random.default_rng(2)
    batches = []
    for i, (nq, nt) in enumerate([(300, 5), (7, 3), (512, 5), (1, 4), (64, 8), (300, 2), (33, 5)]):
        toks = np.stack([rng.choice(33_000, nt, replace=False) for _ in range(nq)]).astype(np.uint32)
        ids = np.sort(seg.token_terms(toks.reshape(-1)).reshape(nq, nt), axis=1).reshape(-1)
        batches.append((ids, (np.arange(nq + 1) * nt).astype(np.uint32)))
    want = [vb.search_batch(gix, t, o, 10) for t, o in batches]
    st = vb.Stream(gix, 3, 512, 4096, 10)
    got = []
    with pytest.raises(vb.Vbm25Error):
        st.collect_raw()
    for i, (t, o) in enumerate(batches):
        if st.in_flight == 3:
            got.append(st.collect())
        st.submit(t, o)
    assert st.in_flight == 3
    with pytest.raises(vb.Vbm25Error):
        st.submit(*batches[0])
    assert len(st._nq) == 3  # (the refused submit never entered the wrapper's bookkeeping: api.py appends only after the library accepted it)
    while st.in_flight:
        got.append(st.collect())
    assert len(got) == len(want)
    for (h, n), (hw, nw) in zip(got, want):
        assert np.array_equal(n, nw) and h.tobytes() == hw.tobytes()


@pytest.mark.parametrize("k", [10, 100])
def test_pipelined_boundary_random_batches_against_the_oracle(k):
    """forty batches of random size (1 .. 300 queries: the one-launch route of a handful of queries and the three-launch routes share a
    ring), random term counts (0 .. 7, unknown tokens among them), through a ring of three slots: every batch's records equal the
    oracle's brute force"""
    import orc
    from parity import assert_bit_exact
    seg = vb.Segment.synth(300_000, 33_000, mean_len=100, len_mode=1, seed=21)
    gix = vb.GpuIndex(seg)
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    rng = np.random.default_rng(k)
    batches = []
    for i in range(40):
        nq = int(rng.choice([1, 2, 5, 8, 9, 33, 100, 300]))
        rows = []
        for q in range(nq):
            nt = int(rng.integers(0, 8))
            ids = seg.token_terms(rng.choice(33_000, nt, replace=False).astype(np.uint32)) if nt else np.zeros(0, dtype=np.uint32)
            if nt and rng.random() < 0.2:
                ids = np.r_[ids[:-1], [0xfffffff0]].astype(np.uint32)  # (a token the index does not know)
            rows.append(np.sort(ids).astype(np.uint32))
        terms = np.concatenate(rows).astype(np.uint32) if rows else np.zeros(0, dtype=np.uint32)
        off = np.r_[0, np.cumsum([len(r) for r in rows])].astype(np.uint32)
        batches.append((terms, off))
    st = vb.Stream(gix, 3, 300, 300 * 8, k)
    got = []
    for t, o in batches:
        if st.in_flight == 3:
            got.append(st.collect())
        st.submit(t, o)
    while st.in_flight:
        got.append(st.collect())
    assert len(got) == len(batches)
    for i, ((t, o), (hits, nh)) in enumerate(zip(batches, got)):
        ob, onb, _ = oix.search_batch(t, o, k, mode="brute", threads=8)
        assert np.array_equal(nh, onb), i
        for q in range(len(o) - 1):
            assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"batch {i} q{q}")
