"""Host side of the shim for the growing (unsealed) segment (include/vbm25.h:
vbm25_growing_search, vbm25_merge_hits) against the oracle's restatement of search.rs:83-135.

The sealed hits fed to the merge come from the oracle here (no GPU needed: the functions under test
are host code); tests/test_gpu_search.py repeats the merge with hits from the device."""
import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus, make_queries
from parity import assert_same_ranking


def _growing(c, seg, rng, n_grow, q_terms):
    """Random unsealed documents: elements = ascending term ranks (some from the query, some
    not), a few keys the index does not hold, random fieldnorms / payloads / deleted flags."""
    a = seg.arrays()
    n_terms = seg.meta()["n_terms"]
    starts, ranks, keys, tfs = [0], [], [], []
    for _ in range(n_grow):
        own = set(rng.choice(n_terms, rng.integers(1, 12), replace=False).tolist())
        own |= set(rng.choice(q_terms, rng.integers(0, len(q_terms) + 1), replace=False).tolist()) if len(q_terms) else set()
        own = sorted(own)
        for r in own:
            ranks.append(r)
            keys.append(a["term_key"][r].tobytes())
            tfs.append(int(rng.integers(1, 6)))
        starts.append(len(ranks))
    fn = rng.integers(0, 120, n_grow).astype(np.uint8)
    payload = rng.integers(0, 60000, (n_grow, 3)).astype(np.uint16)
    deleted = (rng.random(n_grow) < 0.15).astype(np.uint8)
    return (np.array(starts, np.uint64), np.array(ranks, np.uint32),
            np.frombuffer(b"".join(keys), np.uint8), np.array(tfs, np.uint32), fn, payload, deleted)


@pytest.mark.parametrize("k", [1, 5, 40])
def test_growing_search_and_merge_match_the_oracle(k):
    c = make_corpus(4000, 400, seed=21, length="lognormal", mean_len=40)
    seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"],
                           c["post_doc"], c["post_tf"])
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    a = seg.arrays()
    terms, off = make_queries(c, 12, 4, seed=3)
    rng = np.random.default_rng(5)
    for q in range(len(off) - 1):
        t = terms[off[q]:off[q + 1]]
        t = t[t < seg.meta()["n_terms"]]
        g_start, g_rank, g_key, g_tf, g_fn, g_pl, g_del = _growing(c, seg, rng, 60, t)
        query = vb.Query([a["term_key"][r].tobytes() for r in t])
        grow = vb.growing_search(seg, query, k, g_start, g_key, g_tf, g_fn, g_pl, g_del)
        # the growing hits alone: the oracle with an empty posting side is not available, so check
        # them against a direct evaluation through the oracle's Cache::evaluate
        assert (np.diff(grow["score"]) <= 0).all() and (grow["score"] > 0).all()
        for h in grow:
            g = 0xFFFFFFFF - int(h["doc_id"])
            assert not g_del[g] and h["payload"].tolist() == g_pl[g].tolist()
            sel = slice(int(g_start[g]), int(g_start[g + 1]))
            want = 0.0
            m = seg.meta()
            for r, tf in zip(g_rank[sel], g_tf[sel]):
                if r in t:  # Cache::new + Cache::evaluate with the sealed segment's statistics
                    want += orc.lib().orc_cache_evaluate(m["n_docs"], int(a["term_df"][r]), m["k1"], m["b"],
                                                         m["sum_len"] / m["n_docs"], int(g_fn[g]), int(tf))
            assert h["score"] == want
        sealed = oix.search_brute(t, k)
        merged = vb.merge_hits(sealed, grow, k)
        ref = oix.search_wand_growing(t, k, g_start, g_rank, g_tf, g_fn, g_pl, g_del)
        ext = vb.merge_hits(oix.search_brute(t, k + 300), vb.growing_search(seg, query, k + 300, g_start, g_key, g_tf, g_fn, g_pl, g_del), k + 300)
        assert_same_ranking(ref, merged, ref_ext=ext, what=f"q{q} growing+sealed")


def test_growing_search_edge_cases():
    c = make_corpus(500, 100, seed=2, length="fixed", mean_len=20)
    seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"],
                           c["post_doc"], c["post_tf"])
    a = seg.arrays()
    q = vb.Query([a["term_key"][3].tobytes(), a["term_key"][7].tobytes()])
    none = dict(g_start=[0], g_key=np.zeros(0, np.uint8), g_tf=[], g_fieldnorm=[], g_payload=np.zeros((0, 3), np.uint16))
    assert len(vb.growing_search(seg, q, 5, **none)) == 0
    # a document without any query token scores 0 and never enters (threshold 0.0 < 0.0 is false)
    key = a["term_key"][50].tobytes()
    g = vb.growing_search(seg, q, 5, [0, 1], np.frombuffer(key, np.uint8), [3], [10], [[1, 2, 3]])
    assert len(g) == 0
    # a key the sealed segment does not hold is ignored on both sides
    unknown = b"zzzzzzzzzzzzzzz\0"
    q2 = vb.Query(sorted([a["term_key"][3].tobytes(), unknown]))
    keys = np.frombuffer(a["term_key"][3].tobytes() + unknown, np.uint8)
    g = vb.growing_search(seg, q2, 5, [0, 2], keys, [2, 9], [10], [[1, 2, 3]])
    g1 = vb.growing_search(seg, q, 5, [0, 1], keys[:16], [2], [10], [[1, 2, 3]])
    assert len(g) == 1 and g["score"][0] == g1["score"][0]
    # ties: the earlier document stays ahead and a later tie does not displace the k-th
    keys2 = np.frombuffer(a["term_key"][3].tobytes() * 3, np.uint8)
    g = vb.growing_search(seg, q, 2, [0, 1, 2, 3], keys2, [2, 2, 2], [10, 10, 10], [[1, 0, 0], [2, 0, 0], [3, 0, 0]])
    assert g["payload"][:, 0].tolist() == [1, 2]
    with pytest.raises(vb.Vbm25Error):
        vb.growing_search(seg, q, 0, **none)
    assert len(vb.merge_hits(np.zeros(0, vb.HIT_DTYPE), np.zeros(0, vb.HIT_DTYPE), 3)) == 0
