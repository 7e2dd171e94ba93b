"""Pins the oracle to everything the reference's own tests hold for this path
(SURVEY section 8(c)): Score bijection, fieldnorm table, sqllogictest id orderings,
and documents the assumed BinaryHeap model."""
import json
import math
import os
import struct

import numpy as np
import pytest

import orc
from corpus import token_keys  # noqa: F401  (import check)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_score_bijection():
    # crates/score/src/lib.rs:82-112
    L = orc.lib()
    vals = [0.0, -0.0, math.inf, -math.inf, math.nan, -math.nan] + [i * 0.1 for i in range(-100, 100)]
    for v in vals:
        back = L.orc_score_to_f64(L.orc_score_from_f64(v))
        assert struct.pack("<d", back) == struct.pack("<d", v)
    assert L.orc_score_from_f64(0.0) == 0
    assert L.orc_score_from_f64(math.inf) == struct.unpack("<q", struct.pack("<d", math.inf))[0]
    # order preserving over non-NaN values (the reason Score exists)
    xs = sorted([-math.inf, -1e300, -2.5, -1e-300, -0.0, 0.0, 1e-300, 0.1, 2.5, 1e300, math.inf])
    keys = [L.orc_score_from_f64(x) for x in xs]
    assert keys == sorted(keys)


def test_fieldnorm_table_matches_reference():
    # bm25.rs:15-272 (fixture generated from the reference by make_fieldnorm_fixture.py)
    table = json.load(open(os.path.join(GOLD, "fieldnorm_table.json")))
    L = orc.lib()
    assert [L.orc_fieldnorm_to_length(i) for i in range(256)] == table
    if os.path.exists("/root/reference/crates/bm25/src/bm25.rs"):
        import re
        src = open("/root/reference/crates/bm25/src/bm25.rs").read()
        body = src[src.index("FIELDNORM_TO_LENGTH"):]
        body = body[body.index("= [") + 3:body.index("];")]
        assert [int(x.replace("_", "")) for x in re.findall(r"[0-9_]+", body)] == table


def test_length_to_fieldnorm():
    # bm25.rs:278-283: largest index whose table length <= length
    L = orc.lib()
    table = json.load(open(os.path.join(GOLD, "fieldnorm_table.json")))
    for f in range(256):
        assert L.orc_length_to_fieldnorm(table[f]) == f
        if f + 1 < 256 and table[f + 1] - table[f] > 1:
            assert L.orc_length_to_fieldnorm(table[f] + 1) == f
            assert L.orc_length_to_fieldnorm(table[f + 1] - 1) == f
    assert L.orc_length_to_fieldnorm(100) == 57  # 96 <= 100 < 104
    assert L.orc_length_to_fieldnorm(0xffffffff) == 255


def test_bm25_arithmetic():
    # bm25.rs:285-295, 340-358: Cache::evaluate == idf*(k1+1)*tf/(tf+s1)
    L = orc.lib()
    n, df, k1, b, avgdl = 1000, 37, 1.2, 0.75, 93.5
    idf = math.log((n + 1.0) / (df + 0.5))
    assert L.orc_idf(n, df) == idf
    for fn, tf in [(0, 1), (57, 1), (57, 3), (120, 7), (255, 1)]:
        dl = float(L.orc_fieldnorm_to_length(fn))
        s1 = k1 * (1.0 - b + b * dl / avgdl)
        assert L.orc_cache_evaluate(n, df, k1, b, avgdl, fn, tf) == (tf * (idf * (k1 + 1.0))) / (tf + s1)
        assert L.orc_tf(fn, tf, k1, b, avgdl) == (tf * (k1 + 1.0)) / (tf + s1)


def _slt_index(doc_ids, fixture):
    docs = {int(i): toks for i, toks in fixture["docs"].items() if int(i) in doc_ids}
    ids = sorted(docs)  # heap-scan order = id order; dense doc ids 0..n-1
    vocab = sorted({t for toks in docs.values() for t in toks}, key=lambda s: s.encode())
    for t in vocab:
        assert len(t.encode()) < 16  # intern() short path (vector.rs:21-24)
    term_key = np.zeros((len(vocab), 16), dtype=np.uint8)
    for i, t in enumerate(vocab):
        term_key[i, :len(t)] = np.frombuffer(t.encode(), dtype=np.uint8)
    rank = {t: i for i, t in enumerate(vocab)}
    post = {}
    for d, i in enumerate(ids):
        for t in docs[i]:
            post[(rank[t], d)] = post.get((rank[t], d), 0) + 1  # tf = number of positions
    keys = sorted(post)
    term_start = np.zeros(len(vocab) + 1, dtype=np.uint64)
    for r, _ in keys:
        term_start[r + 1] += 1
    term_start = np.cumsum(term_start).astype(np.uint64)
    doc_len = np.array([len(docs[i]) for i in ids], dtype=np.uint32)
    payload = np.array([[0, 0, i] for i in ids], dtype=np.uint16)  # ctid (0, id)
    ix = orc.OracleIndex.build(fixture["k1"], fixture["b"], doc_len, payload, term_key, term_start,
                               np.array([d for _, d in keys], dtype=np.uint32),
                               np.array([post[k] for k in keys], dtype=np.uint32))
    q = np.array(sorted(rank.get(t, len(vocab)) for t in fixture["query"]), dtype=np.uint32)
    return ix, q, ids


def test_sqllogictest_golden_orders():
    fixture = json.load(open(os.path.join(GOLD, "slt_corpus.json")))
    for case in fixture["expect"]:
        sel = {"all": range(1, 11), "even": range(2, 11, 2), "odd": range(1, 11, 2)}[case["ids"]]
        ix, q, ids = _slt_index(set(sel), fixture)
        for fn in (ix.search_wand, ix.search_brute):
            hits = fn(q, case["k"])
            got = [int(h["payload"][2]) for h in hits]
            assert got == case["order"], (case["name"], got)
            assert all(hits["score"][i] > hits["score"][i + 1] for i in range(len(hits) - 1))


def test_heap_model_basic_properties():
    # The BinaryHeap model is ASSUMED (Rust std is not in the reference tree).  These are
    # the properties any correct max-heap has, plus the documented tie behaviours we rely on.
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 50, 200).astype(np.int64)
    ops = np.zeros(200, dtype=np.int32)
    popped, sorted_tags = orc.heap_script(keys, ops)
    assert len(popped) == 0
    assert [keys[t] for t in sorted_tags] == sorted(keys.tolist())  # into_sorted_vec ascending
    # push/pop interleaved: every pop returns the current maximum key
    ops = np.array([0, 0, 0, -1, 0, 0, -1, -1, 0, -1], dtype=np.int32)
    keys = np.array([5, 9, 7, 0, 9, 1, 0, 0, 4, 0], dtype=np.int64)
    popped, rest = orc.heap_script(keys, ops)
    assert [int(keys[t]) for t in popped] == [9, 9, 7, 5]
    assert sorted(int(keys[t]) for t in rest) == [1, 4]
    # sift_up stops on `<=`: an equal key pushed later does not displace the older root
    popped, _ = orc.heap_script(np.array([3, 3, 0], dtype=np.int64),
                                np.array([0, 0, -1], dtype=np.int32))
    assert popped.tolist() == [0]


def test_product_tables_match_the_reference():
    """The PRODUCT's own FIELDNORM_TO_LENGTH (csrc/segment.cpp) against the table generated from bm25.rs:15-272 -- not only the
    oracle's copy -- and its s1[256] (Cache::new, bm25.rs:349-352) bit for bit against the formula evaluated here in f64."""
    import numpy as np
    import vectorchord_bm25_amd as vb
    table = json.load(open(os.path.join(GOLD, "fieldnorm_table.json")))
    t = np.zeros(256, dtype=np.uint32)
    vb._lib.check(vb.lib().vbm25_fieldnorm_table(t.ctypes.data))
    assert t.tolist() == table
    for n_docs, sum_len, k1, b in ((10_000_000, 958_123_457, 1.2, 0.75), (3, 7, 2.0, 0.0), (1000, 10 ** 12, 0.9, 1.0)):
        s1 = np.zeros(256, dtype=np.float64)
        vb._lib.check(vb.lib().vbm25_cache_s1(n_docs, sum_len, k1, b, s1.ctypes.data))
        avgdl = float(sum_len) / float(n_docs)
        want = np.array([k1 * (1.0 - b + b * float(x) / avgdl) for x in table], dtype=np.float64)
        assert s1.view(np.uint64).tolist() == want.view(np.uint64).tolist()
