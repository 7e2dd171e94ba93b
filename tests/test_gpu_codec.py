"""The HIP decoders fed every codec shape of compression.rs:65-136 through small hand-made indexes, against the oracle bit for
bit -- through scan_range_kernel (plan-free and general route), scan_win_kernel (win_force: whatever the lists' lengths; it reads the
window planes derived from the decoded blocks, and the blob in its cold pass), the one-launch route, scan_dense_kernel (every query
declared dense) and the exhaustive scan_many_kernel (k = 300); and through a SECOND index of the same segment made without the
post_id16 and post_rel16 planes (tuning id16_plane = 0, rel16_plane = 0): scan_win_kernel behind decode_id16_kernel, which unpacks
the batch's terms from the blob per launch, and scan_range_kernel's in-kernel decode.  -m gpu only.

* byte-packed TAIL blocks with document-id byte widths 3 and 4 (gaps >= 2^16 and >= 2^24; width 4 is raw absolute ids,
  bytepacking_u32_ordered.rs:200-214 -- its own branch in decode.h) and term-frequency byte widths 2 and 3;
* full bit-packed blocks of EVERY document-id width 1..25 and tf width 1..17 (bitpacking_u32_ordered.rs:222-237);
* widths 26, 27 and 28 on an index of 2^28 documents (a gap of w bits needs that many documents; 29..31 would need the
  per-document arrays of 2^29..2^31 documents -- 5 to 21 GB on the host alone -- and stay with the decode unit tests of the
  oracle and of index creation, which decodes every block of every index it is given)."""
import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from parity import assert_bit_exact

pytestmark = pytest.mark.gpu


def _index(n_docs, lists, seed=0):
    """lists: [(docs ascending, tfs)] one per term -> (segment, arrays, GpuIndex, oracle index)."""
    rng = np.random.default_rng(seed)
    keys = np.zeros((len(lists), 16), dtype=np.uint8)
    for i in range(len(lists)):
        s = b"t%03d" % i
        keys[i, :len(s)] = np.frombuffer(s, dtype=np.uint8)
    term_start = np.cumsum([0] + [len(d) for d, _ in lists]).astype(np.uint64)
    post_doc = np.concatenate([np.asarray(d, dtype=np.uint32) for d, _ in lists])
    post_tf = np.concatenate([np.asarray(t, dtype=np.uint32) for _, t in lists])
    # (2^31 documents: drawn as 32-bit values, the 64-bit ones would be another 17 GB)
    doc_len = rng.integers(1, 3000, n_docs, dtype=np.uint32) if n_docs > (1 << 29) else rng.integers(1, 3000, n_docs).astype(np.uint32)
    seg = vb.Segment.build(1.2, 0.75, doc_len, np.zeros((n_docs, 3), dtype=np.uint16), keys, term_start, post_doc, post_tf)
    a = seg.arrays()
    gix = vb.GpuIndex(seg)
    vb.set_tuning("id16_plane", 0)  # (read at index creation)
    vb.set_tuning("rel16_plane", 0)
    try:
        gix.no_planes = vb.GpuIndex(seg)
    finally:
        vb.reset_tuning()
    assert gix.no_planes.device_bytes < gix.device_bytes
    return seg, a, gix, orc.OracleIndex.from_arrays(seg.meta(), a)


def _check_routes(tuning, gix, oix, terms, off, ks=(10, 128)):
    nq = len(off) - 1
    routes = [("range", dict(win=0, fused=0)), ("range-general", dict(win=0, fused=0, arith=0)), ("win", dict(win_force=1, fused=0)),
              ("fused", dict(fused=1)), ("dense", dict(fused=0, dense_x1000=0)),
              ("win-decode", dict(win_force=1, fused=0)), ("range-decode", dict(win=0, fused=0)), ("fused-decode", dict(fused=1))]
    for name, tune in routes:
        vb.reset_tuning()
        tuning(**tune)
        for k in ks:
            b = vb.Batch(gix.no_planes if name.endswith("-decode") else gix, nq, len(terms), k)
            b.set_queries(terms, off)
            b.run()
            hits, nh = b.fetch()
            ob, onb, _ = oix.search_batch(terms, off, k, mode="brute", threads=8)
            assert np.array_equal(nh, onb), name
            for q in range(nq):
                assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"{name} k={k} q{q}")
    vb.reset_tuning()
    hits, nh = vb.search_batch(gix, terms, off, 300)  # 256 < k: scan_many_kernel
    ob, onb, _ = oix.search_batch(terms, off, 300, mode="brute", threads=8)
    assert np.array_equal(nh, onb)
    for q in range(nq):
        assert_bit_exact(ob[q, :onb[q]], hits[q, :nh[q]], what=f"many q{q}")


def test_byte_packed_tails_of_width_3_and_4(tuning):
    n_docs = (1 << 25) + 5000
    rng = np.random.default_rng(4)
    a_docs = 7 + np.cumsum(rng.integers(1 << 16, 1 << 17, 40))               # tail of 40: gaps >= 2^16 -> 3-byte deltas
    a_tf = rng.integers(1, 60000, 40)
    a_tf[3] = 65535                                                         # 2-byte term frequencies
    b_docs = np.array([5, 5 + (1 << 24) + 3, 5 + (1 << 25) + 9])             # a gap >= 2^24 -> width 4: raw absolute ids
    b_tf = np.array([3, 1 << 17, 70000])                                     # 3-byte term frequencies
    c_docs = np.r_[np.arange(128) * 3 + 1, (1 << 24) + 77]                   # a full block and a tail of ONE posting
    c_tf = np.r_[rng.integers(1, 4, 128), 2]
    d_docs = np.arange(0, n_docs - 1, 100_003)[:336]                         # two full wide blocks (no plane word) + a 3-byte tail of 80
    d_tf = rng.integers(1, 300, len(d_docs))
    e_docs = np.unique(np.r_[a_docs[::2], b_docs, c_docs[::7], d_docs[::5], rng.integers(0, n_docs, 300)])  # meets all of them
    e_tf = rng.integers(1, 5, len(e_docs))
    seg, a, gix, oix = _index(n_docs, [(a_docs, a_tf), (b_docs, b_tf), (c_docs, c_tf), (d_docs, d_tf), (e_docs, e_tf)])
    first = a["term_first_block"]
    md, mt, nblk = a["blk_meta_doc"], a["blk_meta_tf"], a["blk_n"]
    assert md[first[0]] == 0x83 and mt[first[0]] == 0x82 and nblk[first[0]] == 40
    assert md[first[1]] == 0x84 and mt[first[1]] == 0x83 and nblk[first[1]] == 3
    assert nblk[first[2] + 1] == 1 and md[first[2]] < 32
    assert md[first[3] + 2] == 0x83 and nblk[first[3] + 2] == 80 and a["blk_max_doc"][first[3]] - a["blk_min_doc"][first[3]] > 65535
    terms = np.array([0, 1, 2, 3, 4, 0, 4, 1, 4, 2, 4, 3, 4, 0, 1, 2, 3, 4, 0, 3, 1, 2], dtype=np.uint32)
    off = np.array([0, 1, 2, 3, 4, 5, 7, 9, 11, 13, 18, 20, 22], dtype=np.uint32)
    _check_routes(tuning, gix, oix, terms, off)


@pytest.mark.parametrize("seed", [1, 2])
def test_every_bit_width_of_full_blocks(tuning, seed):
    n_docs = (1 << 25) + 200_000
    rng = np.random.default_rng(seed)
    lists = []
    for w in range(1, 26):  # 128 postings: gaps below 2^w, ONE of them with bit w - 1 set
        gaps = rng.integers(1, min(1 << w, 400) + 1, 128) if w > 1 else np.ones(128, dtype=np.int64)
        gaps = np.minimum(gaps, (1 << w) - 1)
        gaps[0] = 0
        gaps[rng.integers(1, 128)] = rng.integers(1 << (w - 1), 1 << w)
        docs = rng.integers(0, 1000) + np.cumsum(gaps)
        wt = 1 + (w - 1) % 17  # tf field width
        tf = rng.integers(1, 1 << min(wt, 3), 128)
        tf[rng.integers(0, 128)] = rng.integers(1 << (wt - 1), 1 << wt)
        lists.append((docs, tf))
    # a term that meets every list a few times, and a dense one
    mix = np.unique(np.concatenate([d[::9] for d, _ in lists] + [rng.integers(0, n_docs, 500)]))
    lists.append((mix, rng.integers(1, 4, len(mix))))
    seg, a, gix, oix = _index(n_docs, lists, seed)
    first = a["term_first_block"]
    for w in range(1, 26):
        assert a["blk_meta_doc"][first[w - 1]] == w and a["blk_meta_tf"][first[w - 1]] == 1 + (w - 1) % 17, w
    nterm = len(lists)
    terms, off = [], [0]
    for t in range(nterm - 1):  # every width alone and with the mixed term
        terms += [t]
        off.append(len(terms))
        terms += [t, nterm - 1]
        off.append(len(terms))
    terms += list(range(0, nterm, 3))  # nine terms at once
    off.append(len(terms))
    terms += list(range(nterm))        # all 26: beyond 16 terms -> scan_many_kernel in every route
    off.append(len(terms))
    _check_routes(tuning, gix, oix, np.array(terms, dtype=np.uint32), np.array(off, dtype=np.uint32), ks=(10,))


def test_bit_widths_26_to_28_of_full_blocks(tuning):
    n_docs = (1 << 28) + 200_000
    rng = np.random.default_rng(7)
    lists = []
    for w in (26, 27, 28):  # 128 postings, ONE gap with bit w - 1 set
        gaps = rng.integers(1, 400, 128)
        gaps[0] = 0
        gaps[rng.integers(1, 128)] = rng.integers(1 << (w - 1), (1 << w) - 60_000)
        docs = rng.integers(0, 1000) + np.cumsum(gaps)
        lists.append((docs, rng.integers(1, 4, 128)))
    mix = np.unique(np.concatenate([d[::7] for d, _ in lists] + [rng.integers(0, n_docs, 300)]))
    lists.append((mix, rng.integers(1, 4, len(mix))))
    seg, a, gix, oix = _index(n_docs, lists, 3)
    first = a["term_first_block"]
    for i, w in enumerate((26, 27, 28)):
        assert a["blk_meta_doc"][first[i]] == w, (w, a["blk_meta_doc"][first[i]])
    terms = np.array([0, 0, 3, 1, 1, 3, 2, 2, 3, 0, 1, 2, 3], dtype=np.uint32)
    off = np.array([0, 1, 3, 4, 6, 7, 9, 13], dtype=np.uint32)
    _check_routes(tuning, gix, oix, terms, off, ks=(10,))


def test_bit_widths_29_to_31_of_full_blocks(tuning):
    """The widest delta fields of bitpacking_u32_ordered.rs:222-237 through every kernel: a gap with bit 30 set (width 31) needs more than
    2^30 documents, so the index has 2^30 + 2^28 of them -- 5.4 GB of document lengths and 8 GB of payload on the host (the GPU box has
    them; this test is -m gpu only), 20480 windows of 2^16 documents on the device."""
    n_docs = (1 << 30) + (1 << 28)
    rng = np.random.default_rng(11)
    lists = []
    for w in (29, 30, 31):  # 128 postings, ONE gap with bit w - 1 set
        gaps = rng.integers(1, 400, 128)
        gaps[0] = 0
        gaps[rng.integers(1, 128)] = rng.integers(1 << (w - 1), min((1 << w), (1 << 30) + (1 << 27)) - 60_000)
        docs = rng.integers(0, 1000) + np.cumsum(gaps)
        assert docs[-1] < n_docs
        lists.append((docs, rng.integers(1, 4, 128)))
    mix = np.unique(np.concatenate([d[::7] for d, _ in lists] + [rng.integers(0, n_docs, 300)]))
    lists.append((mix, rng.integers(1, 4, len(mix))))
    seg, a, gix, oix = _index(n_docs, lists, 5)
    first = a["term_first_block"]
    for i, w in enumerate((29, 30, 31)):
        assert a["blk_meta_doc"][first[i]] == w, (w, a["blk_meta_doc"][first[i]])
    terms = np.array([0, 0, 3, 1, 1, 3, 2, 2, 3, 0, 1, 2, 3], dtype=np.uint32)
    off = np.array([0, 1, 3, 4, 6, 7, 9, 13], dtype=np.uint32)
    _check_routes(tuning, gix, oix, terms, off, ks=(10,))
