"""vbm25_evaluate (host; `tsvector <&> bm25query` as a plain function, evaluate.rs:22-74) against the
oracle's restatement, bit for bit."""
import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus


def test_evaluate_matches_the_oracle_bitwise():
    c = make_corpus(3000, 400, seed=13, length="lognormal", mean_len=40)
    seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"],
                           c["post_doc"], c["post_tf"])
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    a = seg.arrays()
    n_terms = seg.meta()["n_terms"]
    rng = np.random.default_rng(1)
    nonzero = 0
    for _ in range(300):
        d_rank = np.sort(rng.choice(n_terms, int(rng.integers(0, 40)), replace=False)).astype(np.uint32)
        d_tf = rng.integers(1, 2000 if rng.random() < 0.1 else 6, len(d_rank)).astype(np.uint32)
        q_rank = np.sort(rng.choice(n_terms, int(rng.integers(0, 8)), replace=False)).astype(np.uint32)
        if len(d_rank) and len(q_rank) and rng.random() < 0.7:  # make them share terms
            q_rank = np.unique(np.r_[q_rank, rng.choice(d_rank, min(3, len(d_rank)), replace=False)]).astype(np.uint32)
        want = orc.lib().orc_score_to_f64(oix.evaluate(d_rank, d_tf, q_rank))
        got = vb.evaluate(seg, [a["term_key"][r].tobytes() for r in d_rank], d_tf,
                          vb.Query([a["term_key"][r].tobytes() for r in q_rank]))
        assert got == want
        nonzero += got > 0
    assert nonzero > 100
    # keys the index does not hold are skipped on both sides of the merge walk
    unknown = b"zzzzzzzzzzzzzzz\0"
    k3 = a["term_key"][3].tobytes()
    q = vb.Query(sorted([k3, unknown]))
    assert vb.evaluate(seg, [k3], [2], q) == vb.evaluate(seg, [k3], [2], vb.Query([k3])) > 0
    with pytest.raises(vb.Vbm25Error):
        vb.evaluate(seg, [k3, k3], [1, 1], q)  # Document keys must be strictly ascending
