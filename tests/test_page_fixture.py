"""The page reader against a relation that tests/golden/make_page_fixture.py assembled with struct.pack only
(a derivation of the on-disk format independent of both csrc/pages.cpp and oracle/pages.cpp), and the cache
fingerprint (vbm25_pages_fingerprint).  No GPU use."""
import json
import math
import os

import numpy as np

import orc
import vectorchord_bm25_amd as vb

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture():
    raw = open(os.path.join(GOLD, "page_fixture.bin"), "rb").read()
    pages = [raw[i:i + 8192] for i in range(0, len(raw), 8192)]
    return pages, json.load(open(os.path.join(GOLD, "page_fixture.json")))


def test_sealed_segment_of_the_struct_pack_relation():
    pages, exp = _fixture()
    seg = vb.segment_from_pages(pages)
    m, a = seg.meta(), seg.arrays()
    assert (m["n_docs"], m["n_terms"], m["n_blocks"], m["sum_len"]) == (exp["n_docs"], 2, 3, exp["sum_len"])
    assert (m["k1"], m["b"]) == (exp["k1"], exp["b"])
    assert [bytes(k).hex() for k in a["term_key"]] == exp["term_key"]
    assert a["term_df"].tolist() == exp["term_df"]
    assert a["term_first_block"].tolist() == exp["term_first_block"]
    assert [[int(f), int(t)] for f, t in zip(a["term_wand_fn"], a["term_wand_tf"])] == exp["term_wand"]
    assert a["doc_fieldnorm"].tolist() == exp["doc_fieldnorm"]
    assert a["doc_payload"].tolist() == [list(p) for p in exp["doc_payload"]]
    blob = a["blob"].tobytes()
    for j, b in enumerate(exp["blocks"]):
        for f in ("min_doc", "max_doc", "n", "wand_fn", "wand_tf", "meta_doc", "meta_tf"):
            assert int(a["blk_" + f][j]) == b[f], (j, f)
        body = blob[8 * int(a["blk_off8"][j]):8 * int(a["blk_off8"][j + 1])]
        d, t = bytes.fromhex(b["doc_bytes"]), bytes.fromhex(b["tf_bytes"])
        pad = lambda x: x + b"\0" * (-len(x) % 8)
        assert body == pad(d) + pad(t), j
    # the flattened arrays decode to the postings the fixture was made from, and rank like plain BM25
    oix = orc.OracleIndex.from_arrays(m, a)
    got = oix.search_brute(np.array([0, 1], dtype=np.uint32), 5)
    n, avgdl = exp["n_docs"], exp["sum_len"] / exp["n_docs"]
    score = [0.0] * n
    for t, p in enumerate(exp["postings"]):
        idf = math.log(1.0 + (n - len(p["docs"]) + 0.5) / (len(p["docs"]) + 0.5))
        for d, tf in zip(p["docs"], p["tfs"]):
            ln = exp["doc_fieldnorm"][d]  # lengths <= 40: fieldnorm code == length
            score[d] += idf * tf * (exp["k1"] + 1.0) / (tf + exp["k1"] * (1.0 - exp["b"] + exp["b"] * ln / avgdl))
    want = sorted(range(n), key=lambda d: (-score[d], d))[:5]
    assert got["doc_id"].tolist() == want
    assert np.allclose(got["score"], [score[d] for d in want], rtol=1e-12)


def test_growing_segment_with_orphans():
    """A `_2` (+ `_1`) whose insert failed before its `_0` in the middle of the tape, and an unfinished one at
    its end: the reference's state machine (search.rs:94-96) drops both."""
    pages, exp = _fixture()
    g = vb.growing_from_pages(pages)
    e = exp["growing"]
    assert g["g_fieldnorm"].tolist() == e["fieldnorm"]
    assert g["g_deleted"].tolist() == e["deleted"]
    assert g["g_payload"].tolist() == e["payload"]
    keys = g["g_key"].reshape(-1, 16)
    for i, doc in enumerate(e["docs"]):
        s, t = int(g["g_start"][i]), int(g["g_start"][i + 1])
        assert [[bytes(keys[p]).hex(), int(g["g_tf"][p])] for p in range(s, t)] == doc


def test_fingerprint_and_seed():
    pages, exp = _fixture()
    assert vb.api.pages_seed(pages).hex() == exp["seed"]
    f0 = vb.api.pages_fingerprint(pages)
    assert len(f0) == 32 and f0 == vb.api.pages_fingerprint(pages)
    # an insert appends to the vectors tape only: same fingerprint
    ins = list(pages)
    ins[7] = ins[7][:100] + b"\x01" + ins[7][101:]
    assert vb.api.pages_fingerprint(ins) == f0
    # VACUUM rewrites the Jump tuple (maintain.rs:268-298): e.g. a new document count / tape pointer
    jump = bytearray(pages[1])
    upper = int.from_bytes(jump[14:16], "little")
    jump[upper + 4] ^= 1
    assert vb.api.pages_fingerprint([pages[0], bytes(jump)] + pages[2:]) != f0
    # REINDEX writes a new Meta tuple (new seed)
    meta = bytearray(pages[0])
    upper = int.from_bytes(meta[14:16], "little")
    meta[upper + 40] ^= 0x80
    assert vb.api.pages_fingerprint([bytes(meta)] + pages[1:]) != f0
