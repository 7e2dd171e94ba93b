"""Faithful Block-WAND restatement vs canonical brute force (the reference's own parity
statement, tests/fuzz:217-303: index scan vs seq scan, top-100, edit distance <= 10)."""
import numpy as np
import pytest

import orc
from corpus import make_corpus, make_queries
from parity import assert_same_ranking, edit_distance


def build(c, k1=1.2, b=0.75):
    return orc.OracleIndex.build(k1, b, c["doc_len"], c["doc_payload"], c["term_key"],
                                 c["term_start"], c["post_doc"], c["post_tf"])


@pytest.fixture(scope="module")
def fuzz_corpus():
    # tests/fuzz:43-45: 10 000 docs, 100 draws each from 10 000 uniform tokens
    c = make_corpus(10000, 10000, seed=11, length="fixed", mean_len=100)
    return c, build(c)


def test_fuzz_rule_wand_vs_brute(fuzz_corpus):
    c, ix = fuzz_corpus
    # queries are random tsvectors too (~100 distinct tokens), top-100
    terms, off = make_queries(c, 20, 100, seed=5)
    for q in range(len(off) - 1):
        t = terms[off[q]:off[q + 1]]
        w = ix.search_wand(t, 100)
        b = ix.search_brute(t, 100)
        assert edit_distance(w["doc_id"].tolist(), b["doc_id"].tolist()) <= 10
        assert_same_ranking(b, w, ref_ext=ix.search_brute(t, 400), what=f"q{q}")


@pytest.mark.parametrize("length,zipf,nterms,k", [
    ("fixed", None, 3, 10), ("lognormal", None, 5, 10), ("mixed", None, 2, 1),
    ("lognormal", 1.0, 10, 100), ("lognormal", 1.0, 4, 7)])
def test_wand_equals_brute_outside_ties(length, zipf, nterms, k):
    c = make_corpus(20000, 2000, seed=7, length=length, mean_len=60, zipf=zipf)
    ix = build(c)
    terms, off = make_queries(c, 40, nterms, seed=9, zipf=zipf)
    for q in range(len(off) - 1):
        t = terms[off[q]:off[q + 1]]
        assert_same_ranking(ix.search_brute(t, k), ix.search_wand(t, k),
                            ref_ext=ix.search_brute(t, k + 500), what=f"q{q}")


def test_edge_cases():
    c = make_corpus(1000, 1000, seed=3, length="fixed", mean_len=100)  # BASELINE config C1
    ix = build(c)
    nt = ix.n_terms
    assert len(ix.search_wand([], 10)) == 0 and len(ix.search_brute([], 10)) == 0
    # unknown tokens are ignored (search.rs:59-61)
    assert len(ix.search_wand([nt + 5], 10)) == 0
    a = ix.search_wand([3, nt + 5], 10)
    b = ix.search_wand([3], 10)
    assert np.array_equal(a["doc_id"], b["doc_id"])
    # k larger than the number of matching docs
    df = int(ix.arrays["term_df"][3])
    assert len(ix.search_wand([3], 5000)) == df == len(ix.search_brute([3], 5000))
    # single-posting term / every term of a document
    t = np.flatnonzero(ix.arrays["term_df"] == ix.arrays["term_df"].min())[:1].astype(np.uint32)
    assert len(ix.search_wand(t, 10)) == min(10, int(ix.arrays["term_df"][t[0]]))


def test_config_c1_three_term_top10():
    c = make_corpus(1000, 1000, seed=20260925 % 2**31, length="lognormal", mean_len=100)
    ix = build(c)
    terms, off = make_queries(c, 50, 3, seed=1)
    out_w, nh_w, _ = ix.search_batch(terms, off, 10, mode="wand", threads=2)
    out_b, nh_b, _ = ix.search_batch(terms, off, 10, mode="brute", threads=2)
    assert np.array_equal(nh_w, nh_b)
    for q in range(50):
        t = terms[off[q]:off[q + 1]]
        assert_same_ranking(out_b[q, :nh_b[q]], out_w[q, :nh_w[q]],
                            ref_ext=ix.search_brute(t, 200), what=f"q{q}")


def test_evaluate_matches_brute_score():
    # bm25::evaluate (the `<&>` seq-scan function) uses idf*tf instead of Cache: same value
    # up to rounding, per document
    c = make_corpus(500, 200, seed=2, length="lognormal", mean_len=40)
    ix = build(c)
    terms, off = make_queries(c, 5, 4, seed=3)
    # rebuild document 0..20 term lists from postings
    ts, pd, ptf = c["term_start"], c["post_doc"], c["post_tf"]
    term_of_post = np.repeat(np.arange(len(ts) - 1), np.diff(ts).astype(np.int64))
    L = orc.lib()
    for q in range(5):
        t = terms[off[q]:off[q + 1]]
        hits = ix.search_brute(t, 500)
        for h in hits[:20]:
            d = int(h["doc_id"])
            sel = pd == d
            key = ix.evaluate(term_of_post[sel], ptf[sel], t)
            assert abs(L.orc_score_to_f64(key) - h["score"]) <= 1e-12 * h["score"]


def test_growing_segment_seeds_threshold():
    c = make_corpus(2000, 300, seed=4, length="lognormal", mean_len=50)
    ix = build(c)
    terms, off = make_queries(c, 1, 3, seed=8)
    t = terms[off[0]:off[1]]
    base = ix.search_wand(t, 5)
    # one growing doc containing every query term with high tf and tiny fieldnorm: must rank first
    g = ix.search_wand_growing(t, 5, [0, len(t)], t, np.full(len(t), 9), [1], [[7, 7, 7]], [0])
    assert g[0]["payload"].tolist() == [7, 7, 7] and g[0]["score"] > base[0]["score"]
    assert np.array_equal(g["doc_id"][1:], base["doc_id"][:4])
    # deleted growing docs are skipped (search.rs:113)
    g2 = ix.search_wand_growing(t, 5, [0, len(t)], t, np.full(len(t), 9), [1], [[7, 7, 7]], [1])
    assert np.array_equal(g2["doc_id"], base["doc_id"])
