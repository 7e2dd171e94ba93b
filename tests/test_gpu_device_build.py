"""Device-resident build: the sealed segment encoded (or generated) in HBM and indexed there (vbm25_device_segment_*,
vbm25_index_create_from_device) -- against the host builder, the oracle's flush and the index made of host arrays.  -m gpu."""
import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus, make_queries
from parity import assert_bit_exact
from test_segment_builder import assert_same_index, decode_all

pytestmark = pytest.mark.gpu


def _same_segment(a, b):
    aa, bb = a.arrays(), b.arrays()
    assert a.meta() == b.meta()
    for name in aa:
        assert np.array_equal(aa[name], bb[name]), name


def test_device_segment_build_is_the_host_builders_segment_and_the_same_index():
    c = make_corpus(150_000, 2500, seed=21, length="lognormal", mean_len=70)
    args = (1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
    host = vb.Segment.build(*args)
    dseg = vb.DeviceSegment.build(*args)
    assert (dseg.n_docs, dseg.n_terms, dseg.n_blocks) == (host.n_docs, host.desc.n_terms, host.desc.n_blocks)
    _same_segment(dseg.download(), host)
    terms, off = make_queries(c, 64, 4, seed=2)
    assert sum(dseg.query_bytes(terms[off[q]:off[q + 1]], 10) for q in range(64)) == \
        sum(host.query_bytes(terms[off[q]:off[q + 1]], 10) for q in range(64))
    # the index made where the segment lies == the index made of the host arrays: same records
    g_dev, g_host = vb.GpuIndex(dseg), vb.GpuIndex(host)
    assert g_dev.device_bytes == g_host.device_bytes
    for k in (10, 300):
        h1, n1 = vb.search_batch(g_dev, terms, off, k)
        h2, n2 = vb.search_batch(g_host, terms, off, k)
        assert np.array_equal(n1, n2) and h1.tobytes() == h2.tobytes()
    del dseg  # the index does not borrow from the segment
    h3, n3 = vb.search_batch(g_dev, terms, off, 10)
    h4, n4 = vb.search_batch(g_host, terms, off, 10)
    assert h3.tobytes() == h4.tobytes()


@pytest.mark.parametrize("len_mode,zipf", [(0, 0.0), (1, 0.0), (1, 1.0)])
def test_device_generated_corpus_is_a_valid_flush(len_mode, zipf):
    """vbm25_device_segment_synth: the model of vbm25_segment_synth run on the device.  The result is a valid sealed segment --
    decoded with the oracle's codec and flushed again by the oracle it is the same arrays --, has the statistics asked for,
    is deterministic, and searches bit-exactly against the oracle through the index made in place."""
    n_docs, vocab, mean = 200_000, 1500, 60
    dseg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean, len_mode=len_mode, zipf_s=zipf, seed=7)
    seg = dseg.download()
    a = seg.arrays()
    docs, tfs, ts = decode_all(a)
    lens = np.zeros(seg.n_docs, dtype=np.int64)
    np.add.at(lens, docs, tfs)
    assert lens.sum() == seg.desc.sum_len and dseg.n_postings == len(docs)
    L = orc.lib()
    fn = np.array([L.orc_length_to_fieldnorm(int(x)) for x in lens], dtype=np.uint8)
    assert np.array_equal(fn, a["doc_fieldnorm"])
    oix = orc.OracleIndex.build(1.2, 0.75, lens.astype(np.uint32), a["doc_payload"], a["term_key"], ts, docs, tfs)
    assert_same_index(seg, oix)
    # statistics of the model: document lengths, token frequencies
    if len_mode == 0 and zipf == 0.0:
        assert abs(lens.mean() / mean - 1) < 0.02  # (a slot is hit by 1.0 tokens on average)
    df = a["term_df"].astype(np.float64)
    if zipf == 0.0:
        assert abs(df.mean() * len(df) / len(docs) - 1) < 1e-9 and df.std() / df.mean() < 0.1
    else:
        toks = np.array([int(bytes(k).rstrip(b"\0")) for k in a["term_key"]])
        assert df[np.argsort(toks)][:5].min() > 20 * np.median(df)  # the head of the Zipf law
    keys = [bytes(k).rstrip(b"\0") for k in a["term_key"]]
    assert keys == sorted(keys) and all(k.isdigit() for k in keys)
    toks = np.array([int(k) for k in keys], dtype=np.uint32)
    assert np.array_equal(dseg.token_terms(toks), np.arange(len(keys), dtype=np.uint32))
    assert np.array_equal(seg.token_terms(toks), np.arange(len(keys), dtype=np.uint32))
    # the host generator makes the same corpus up to libm / ocml rounding: the same model, nearly the same postings.  (Zipf: the
    # device draws a head token's chunk as several independent parts -- other streams, the same law: equal within sampling noise)
    host = vb.Segment.synth(n_docs, vocab, mean_len=mean, len_mode=len_mode, zipf_s=zipf, seed=7, threads=4)
    tol = max(50, len(docs) // 10_000) if zipf == 0.0 else len(docs) // 500
    assert abs(int(host.arrays()["term_df"].astype(np.int64).sum()) - len(docs)) <= tol
    # deterministic
    again = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean, len_mode=len_mode, zipf_s=zipf, seed=7).download()
    assert np.array_equal(again.arrays()["blob"], a["blob"])
    # searched where it was made
    gix = vb.GpuIndex(dseg)
    rng = np.random.default_rng(3)
    nq = 48
    tok = np.stack([rng.choice(vocab, 4, replace=False) for _ in range(nq)]).astype(np.uint32)
    terms = np.sort(dseg.token_terms(tok.reshape(-1)).reshape(nq, 4), axis=1).reshape(-1)
    off = (np.arange(nq + 1) * 4).astype(np.uint32)
    hits, nh = vb.search_batch(gix, terms, off, 10)
    ob, onb, _ = oix.search_batch(terms, off, 10, mode="brute", threads=8)
    assert np.array_equal(nh, onb)
    for q in range(nq):
        assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"q{q}")
