"""The scheme of the device's dense-window kernel (scan_dense_kernel) checked on the CPU through its scalar
model (oracle/dense_model.inc): 16-bit fixed-point upper-bound sums select, exact f64 sums decide -- the hits must be those
of the canonical brute force for every window size, with and without the MaxScore split / block skipping,
and the skipping must actually skip on a Zipf corpus."""
import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus, make_queries


def oracle_of(c):
    seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"],
                           c["post_doc"], c["post_tf"])
    return orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())


@pytest.mark.parametrize("zipf,nterms,k", [(1.0, 10, 100), (1.0, 4, 7), (None, 5, 10), (1.0, 16, 256)])
def test_model_matches_brute_force(zipf, nterms, k):
    c = make_corpus(40000, 3000, seed=7, length="lognormal", mean_len=60, zipf=zipf)
    oix = oracle_of(c)
    terms, off = make_queries(c, 12, nterms, seed=9, zipf=zipf)
    for q in range(len(off) - 1):
        t = terms[off[q]:off[q + 1]]
        ref = oix.search_brute(t, k)
        for wmax, w0, ph in ((16384, 256, 1), (1024, 64, 2), (8192, 0, 0), (512, 512, 1), (4096, 256, 3)):
            got, st = oix.dense_model(t, k, wmax=wmax, w0=w0, phases=ph)
            assert got.tobytes() == ref.tobytes(), (q, wmax, w0, ph)


def test_model_on_sub_ranges_and_ties():
    # identical documents: every score ties; doc-range items must keep the canonical (score desc, id asc) order
    n = 5000
    keys = np.zeros((2, 16), dtype=np.uint8)
    keys[:, 0] = [ord("a"), ord("b")]
    docs = np.arange(n, dtype=np.uint32)
    seg = vb.Segment.build(1.2, 0.75, np.full(n, 10, dtype=np.uint32), np.zeros((n, 3), dtype=np.uint16), keys,
                           np.array([0, n, 2 * n], dtype=np.uint64), np.r_[docs, docs], np.ones(2 * n, dtype=np.uint32))
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    t = np.array([0, 1], dtype=np.uint32)
    ref = oix.search_brute(t, 300)
    got, _ = oix.dense_model(t, 300, wmax=1024, w0=64)
    assert got.tobytes() == ref.tobytes()
    got, _ = oix.dense_model(t, 50, wmax=1024, w0=64, lo=1000, hi=3000)
    assert list(got["doc_id"]) == list(range(1000, 1050))


def test_block_skipping_skips_on_zipf():
    seg = vb.Segment.synth(300_000, 30_000, mean_len=100, len_mode=1, zipf_s=1.0, seed=3)
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_queries as bench_queries
    terms, off = bench_queries(seg, 30_000, 6, 10, seed=1, zipf_s=1.0)
    tested = skipped = cand = resc = 0
    for q in range(6):
        t = terms[off[q]:off[q + 1]]
        got, st = oix.dense_model(t, 100, wmax=16384, w0=256, phases=1)
        assert got.tobytes() == oix.search_brute(t, 100).tobytes()
        tested += st["tested"]
        skipped += st["skipped"]
        cand += st["candidates"]
        resc += st["rescored"]
    assert tested > 0 and skipped > 0.3 * tested, (tested, skipped)
    assert resc < cand, (cand, resc)  # the flush drops candidates the threshold has overtaken
