"""vbm25_segment_build_device (csrc/flush.hip: flush.rs:40-158 on the GPU) against the host builder and the
oracle's restatement of flush: every output array byte for byte.  -m gpu only."""
import time

import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus

pytestmark = pytest.mark.gpu


def assert_same_segment(dev, host, oix=None):
    a, h = dev.arrays(), host.arrays()
    assert dev.meta() == host.meta()
    for name in h:
        assert np.array_equal(a[name].reshape(-1), h[name].reshape(-1)), name
    if oix is not None:
        for name in oix.arrays:
            assert np.array_equal(a[name].reshape(-1), oix.arrays[name].reshape(-1)), name


@pytest.mark.parametrize("n_docs,vocab,length,zipf", [
    (1000, 1000, "fixed", None), (3000, 200, "lognormal", None), (5000, 50, "mixed", 1.0),
    (700, 3, "fixed", None), (200_000, 5000, "lognormal", 1.0)])
def test_device_flush_is_byte_identical(n_docs, vocab, length, zipf):
    c = make_corpus(n_docs, vocab, seed=n_docs, length=length, mean_len=40, zipf=zipf)
    args = (c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
    oix = orc.OracleIndex.build(1.2, 0.75, *args) if n_docs <= 5000 else None
    t0 = time.perf_counter()
    host = vb.Segment.build(1.2, 0.75, *args)
    t1 = time.perf_counter()
    dev = vb.Segment.build_device(1.2, 0.75, *args)
    t2 = time.perf_counter()
    assert_same_segment(dev, host, oix)
    print(f"{len(c['post_doc'])} postings: host builder {t1 - t0:.3f} s, device builder {t2 - t1:.3f} s (incl. PCIe both ways)")
    assert_same_segment(vb.Segment.build_device(2.0, 0.0, *args), vb.Segment.build(2.0, 0.0, *args))
    # and the segment serves searches like any other
    gix = vb.GpuIndex(dev)
    t = np.array([0, 1], dtype=np.uint32)
    hits, nh = vb.search_batch(gix, t, np.array([0, 2], dtype=np.uint32), 5)
    want = orc.OracleIndex.from_arrays(dev.meta(), dev.arrays()).search_brute(t, 5)
    assert np.array_equal(hits[0, :nh[0]]["doc_id"], want["doc_id"])


def test_device_flush_codec_corner_cases():
    # bitwidth-32 raw block, df == 128 exactly, a single posting, a 4-byte tf (as in test_segment_builder)
    n_docs = 3_000_000
    docs_a = np.r_[np.arange(64), 2_900_000 + np.arange(64) * 3].astype(np.uint32)
    docs_b = (np.arange(128) * 7 + 5).astype(np.uint32)
    docs_c = np.array([123456], dtype=np.uint32)
    docs_d = (np.arange(300) * 9000 + 17).astype(np.uint32)
    post_doc = np.r_[docs_a, docs_b, docs_c, docs_d]
    rng = np.random.default_rng(0)
    post_tf = np.r_[np.ones(128), rng.integers(1, 70000, 128), [1 << 30], rng.integers(1, 4, 300)].astype(np.uint32)
    term_start = np.array([0, 128, 256, 257, 557], dtype=np.uint64)
    keys = np.zeros((4, 16), dtype=np.uint8)
    keys[:, 0] = [ord("a"), ord("b"), ord("c"), ord("d")]
    args = (np.full(n_docs, 10, dtype=np.uint32), np.zeros((n_docs, 3), dtype=np.uint16), keys, term_start, post_doc, post_tf)
    dev = vb.Segment.build_device(1.2, 0.75, *args)
    assert_same_segment(dev, vb.Segment.build(1.2, 0.75, *args), orc.OracleIndex.build(1.2, 0.75, *args))
    a = dev.arrays()
    assert a["blk_meta_doc"][0] == 22 and a["blk_n"].tolist() == [128, 128, 1, 128, 128, 44]
    assert a["blk_meta_tf"][2] == 0x84


def test_device_flush_rejects_bad_mappings():
    c = make_corpus(500, 50, seed=2, length="fixed", mean_len=20)
    bad = c["post_doc"].copy()
    bad[5], bad[6] = bad[6], bad[5]  # not sorted inside a term
    with pytest.raises(vb.Vbm25Error) as e:
        vb.Segment.build_device(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], bad, c["post_tf"])
    assert e.value.code == -1


def _triples(c):
    """The CSR mappings of a corpus as (token rank, document, tf) triples."""
    ts = c["term_start"].astype(np.int64)
    term = np.repeat(np.arange(len(ts) - 1, dtype=np.uint32), np.diff(ts))
    return term, c["post_doc"].astype(np.uint32), c["post_tf"].astype(np.uint32)


@pytest.mark.parametrize("n_docs,vocab,length,zipf", [
    (1000, 1000, "fixed", None), (5000, 50, "mixed", 1.0), (700, 3, "fixed", None), (200_000, 5000, "lognormal", 1.0)])
def test_device_build_from_unsorted_mappings_is_byte_identical(n_docs, vocab, length, zipf):
    """vbm25_segment_build_device_unsorted: the triples shuffled (the order a tokenizer or several writers would hand them
    over in) -> radix sort on the device into segment.rs:41-45's (token, document) order -> the device encode; every array
    byte for byte what the host builder makes of the sorted CSR form."""
    c = make_corpus(n_docs, vocab, seed=n_docs + 1, length=length, mean_len=40, zipf=zipf)
    host = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"], c["post_doc"], c["post_tf"])
    term, doc, tf = _triples(c)
    perm = np.random.default_rng(5).permutation(len(term))
    t0 = time.perf_counter()
    dev = vb.Segment.build_device_unsorted(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], term[perm], doc[perm], tf[perm])
    print(f"{len(term)} mappings sorted and encoded on the device in {time.perf_counter() - t0:.3f} s (incl. PCIe both ways)")
    assert_same_segment(dev, host)
    # already sorted input and reversed input give the same bytes
    assert_same_segment(vb.Segment.build_device_unsorted(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], term[::-1], doc[::-1], tf[::-1]), host)


def test_device_build_from_unsorted_mappings_rejects_bad_input():
    c = make_corpus(500, 50, seed=2, length="fixed", mean_len=20)
    term, doc, tf = _triples(c)
    args = (1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"])
    with pytest.raises(vb.Vbm25Error):  # a token rank beyond the keys
        vb.Segment.build_device_unsorted(*args, np.r_[term, np.uint32(len(c["term_key"]))], np.r_[doc, np.uint32(0)], np.r_[tf, np.uint32(1)])
    for far in (len(c["term_key"]) + 1, 0x7fffffff, 0xffffffff):  # far beyond them (term_start has n_terms + 1 entries)
        with pytest.raises(vb.Vbm25Error) as e:
            vb.Segment.build_device_unsorted(*args, np.r_[term, np.uint32(far)], np.r_[doc, np.uint32(0)], np.r_[tf, np.uint32(1)])
        assert e.value.code == -1
    # ... and the device is still usable afterwards
    vb.Segment.build_device_unsorted(*args, term, doc, tf)
    with pytest.raises(vb.Vbm25Error):  # the same (token, document) twice
        vb.Segment.build_device_unsorted(*args, np.r_[term, term[:1]], np.r_[doc, doc[:1]], np.r_[tf, tf[:1]])
    with pytest.raises(vb.Vbm25Error):  # a token without mappings
        keep = term != 7
        vb.Segment.build_device_unsorted(*args, term[keep], doc[keep], tf[keep])
    with pytest.raises(vb.Vbm25Error):  # tf = 0
        z = tf.copy()
        z[3] = 0
        vb.Segment.build_device_unsorted(*args, term, doc, z)
