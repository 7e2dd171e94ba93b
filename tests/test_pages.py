"""The product's page reader (include/vbm25.h: vbm25_segment_from_pages / vbm25_growing_from_pages)
against relations laid out in the reference's on-disk format by oracle/pages.cpp (restatement of
build.rs / flush.rs / tape.rs / tuples.rs / insert.rs over PostgreSQL pages).

What pins the writer (no reference page image is available here): the per-page tuple counts the
reference's format implies -- 226 TokenTuples, 291 SummaryTuples, 680 DocumentTuples per 8 KiB page,
2036 / 407 entries per address page."""
import struct

import numpy as np
import pytest

import orc
import vectorchord_bm25_amd as vb
from corpus import make_corpus


def _relation(n_docs=3000, vocab=500, seed=3):
    c = make_corpus(n_docs, vocab, seed=seed, length="lognormal", mean_len=40)
    seg = vb.Segment.build(1.2, 0.75, c["doc_len"], c["doc_payload"], c["term_key"], c["term_start"],
                           c["post_doc"], c["post_tf"])
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    return c, seg, oix, orc.Pages(oix, seed=bytes(range(32)))


def _page_list(pages):
    return [pages.page(i) for i in range(len(pages))]


def _slots(page):
    lower = struct.unpack_from("<H", page, 12)[0]
    out = []
    for i in range((lower - 24) // 4):
        iid = struct.unpack_from("<I", page, 24 + 4 * i)[0]
        out.append((iid & 0x7fff, iid >> 17))
    return out


def _next(page):
    return struct.unpack_from("<I", page, 8192 - 8)[0]


def test_writer_layout_known_answers():
    c, seg, oix, pages = _relation()
    pl = _page_list(pages)
    meta = bytes(pl[0])
    (off, size), = _slots(pl[0])
    assert size == 72 and meta[off:off + 8] == b"vchordbm" and struct.unpack_from("<Q", meta, off + 8)[0] == 1
    k1, b, ptr_lock, ptr_jump = struct.unpack_from("<ddII", meta, off + 16)
    assert (k1, b) == (1.2, 0.75) and meta[off + 40:off + 72] == bytes(range(32))
    joff, jsize = _slots(pl[ptr_jump])[0]
    assert jsize == 64
    j = struct.unpack_from("<IIQHHIIIIIIIIII", bytes(pl[ptr_jump]), joff)
    (ptr_vectors, n_docs, sum_len, w1, w0, depth_d, start_d, free_d, depth_t, start_t, free_t,
     ptr_documents, ptr_tokens, ptr_summaries, ptr_blocks) = j
    assert n_docs == 3000 and sum_len == seg.meta()["sum_len"]
    assert (w1, w0) == (2036, 680)                       # address_documents widths
    assert depth_d == 1 and depth_t == 1                   # 5 document pages, 3 token pages: one level each
    # tuples per page
    counts = []
    p = ptr_documents
    while p != 0xFFFFFFFF:
        counts.append(len(_slots(pl[p])))
        p = _next(pl[p])
    assert counts[:-1] == [680] * (len(counts) - 1) and sum(counts) == 3000
    p, counts = ptr_tokens, []
    while p != 0xFFFFFFFF:
        counts.append(len(_slots(pl[p])))
        p = _next(pl[p])
    assert counts[:-1] == [226] * (len(counts) - 1) and sum(counts) == seg.meta()["n_terms"]
    p, counts = ptr_summaries, []
    while p != 0xFFFFFFFF:
        counts.append(len(_slots(pl[p])))
        p = _next(pl[p])
    assert counts[:-1] == [291] * (len(counts) - 1) and sum(counts) == seg.meta()["n_blocks"]
    # every tuple is 8-byte aligned and pages never overlap their line pointer array
    for pg in pl:
        lower, upper = struct.unpack_from("<HH", pg, 12)
        assert 24 <= lower <= upper <= 8184
        for off, size in _slots(pg):
            assert off % 8 == 0 and off >= upper and off + size <= 8184
    # build.rs allocation order: meta 0, documents tape 1, ... vectors, jump, lock last
    assert ptr_documents == 1 and ptr_jump == ptr_vectors + 1 and ptr_lock == ptr_jump + 1 == len(pl) - 1


def test_sealed_segment_round_trip_is_byte_identical():
    c, seg, oix, pages = _relation()
    flat = vb.segment_from_pages(_page_list(pages))
    m0, m1 = seg.meta(), flat.meta()
    assert m0 == m1
    a0, a1 = seg.arrays(), flat.arrays()
    for name in a0:
        assert np.array_equal(a0[name], a1[name]), name
    # also through the address-returning callable (no copies of the pages)
    flat2 = vb.segment_from_pages(lambda i: pages.address(i))
    assert np.array_equal(flat2.arrays()["blob"], a0["blob"])


def test_tail_blocks_and_many_pages():
    # df % 128 != 0 everywhere (byte-packed tails), blocks spread over many pages
    c, seg, oix, pages = _relation(n_docs=20000, vocab=300, seed=9)
    assert len(pages) > 100
    flat = vb.segment_from_pages(_page_list(pages))
    a0, a1 = seg.arrays(), flat.arrays()
    for name in a0:
        assert np.array_equal(a0[name], a1[name]), name


def test_growing_segment_round_trip():
    c, seg, oix, pages = _relation()
    a = seg.arrays()
    rng = np.random.default_rng(4)
    docs = []
    for i in range(120):
        # some documents are far larger than a page (continuation tuples, page moves), some empty
        n = int(rng.choice([0, 1, 3, 40, 200, 900]))
        ranks = np.sort(rng.choice(seg.meta()["n_terms"], min(n, seg.meta()["n_terms"]), replace=False))
        keys = [a["term_key"][r].tobytes() for r in ranks]
        tfs = rng.integers(1, 9, len(keys)).astype(np.uint32)
        payload = rng.integers(0, 65535, 3).astype(np.uint16)
        pages.insert(payload, keys, tfs)
        docs.append((keys, tfs, payload))
    pages.mark_deleted_growing(5)
    pages.mark_deleted_growing(77)
    g = vb.growing_from_pages(_page_list(pages))
    assert len(g["g_start"]) == 121 and g["g_deleted"].sum() == 2 and g["g_deleted"][5] and g["g_deleted"][77]
    for i, (keys, tfs, payload) in enumerate(docs):
        s, e = int(g["g_start"][i]), int(g["g_start"][i + 1])
        assert e - s == len(keys)
        assert g["g_key"][16 * s:16 * e].tobytes() == b"".join(keys)
        assert np.array_equal(g["g_tf"][s:e], tfs)
        assert g["g_payload"][i].tolist() == payload.tolist()
        want_fn = orc.lib().orc_length_to_fieldnorm(int(min(int(tfs.sum()), 0xFFFFFFFF)))
        assert g["g_fieldnorm"][i] == want_fn
    # the sealed part is untouched by inserts
    flat = vb.segment_from_pages(_page_list(pages))
    assert np.array_equal(flat.arrays()["blob"], a["blob"])
    # and the whole shim path on the host: growing hits from pages == growing hits from the arrays
    q = vb.Query([a["term_key"][r].tobytes() for r in (3, 50, 200)])
    h = vb.growing_search(flat, q, 10, **g)
    assert len(h) > 0 and (np.diff(h["score"]) <= 0).all()


def test_corruption_is_reported_not_crashed():
    c, seg, oix, pages = _relation(n_docs=800, vocab=100)
    pl = _page_list(pages)

    def broken(mutate):
        cp = [p.copy() for p in pl]
        mutate(cp)
        with pytest.raises(vb.Vbm25Error) as e:
            vb.segment_from_pages(cp)
        assert e.value.code == -2 and "data corruption" in str(e.value)

    off = _slots(pl[0])[0][0]
    broken(lambda cp: cp[0].__setitem__(slice(off, off + 8), np.frombuffer(b"notmagic", np.uint8)))
    broken(lambda cp: cp[0].__setitem__(off + 8, 2))                      # version
    broken(lambda cp: cp[1].__setitem__(slice(8184, 8188), np.frombuffer(struct.pack("<I", 10**6), np.uint8)))  # next -> nowhere
    broken(lambda cp: cp[1].__setitem__(slice(12, 14), np.frombuffer(struct.pack("<H", 9000), np.uint8)))      # pd_lower
    # a summary whose block pointer is off by one slot
    ptr_jump = struct.unpack_from("<I", bytes(pl[0]), off + 36)[0]
    joff = _slots(pl[ptr_jump])[0][0]
    ptr_summaries = struct.unpack_from("<I", bytes(pl[ptr_jump]), joff + 52)[0]
    soff = _slots(pl[ptr_summaries])[0][0]
    broken(lambda cp: cp[ptr_summaries].__setitem__(soff + 12, cp[ptr_summaries][soff + 12] + 1))
    with pytest.raises(vb.Vbm25Error):
        vb.segment_from_pages([])


def test_random_damage_never_crashes_the_reader():
    """Byte flips anywhere in the relation: the reader returns a segment or VBM25_ERR_CORRUPT, and a
    segment it returns passes the library's own structural checks or is rejected by them -- it never
    reads outside the 8 KiB images (the process would die)."""
    c, seg, oix, pages = _relation(n_docs=1500, vocab=120, seed=5)
    a = seg.arrays()
    for i in range(6):
        pages.insert(np.array([i, 1, 2], np.uint16), [a["term_key"][r].tobytes() for r in (1, 5, 9)], [1, 2, 3])
    pl = _page_list(pages)
    rng = np.random.default_rng(0)
    outcomes = {"ok": 0, "corrupt": 0}
    for it in range(300):
        cp = [p.copy() for p in pl]
        for _ in range(int(rng.integers(1, 4))):
            pg = int(rng.integers(0, len(cp)))
            if rng.random() < 0.5:   # header / line pointers / special area
                pos = int(rng.choice(np.r_[np.arange(12, 60), np.arange(8184, 8192)]))
            else:
                pos = int(rng.integers(0, 8192))
            cp[pg][pos] = int(rng.integers(0, 256))
        for fn in (vb.segment_from_pages, vb.growing_from_pages):
            try:
                fn(cp)
                outcomes["ok"] += 1
            except vb.Vbm25Error as e:
                assert e.code in (-2, -1)
                outcomes["corrupt"] += 1
    assert outcomes["ok"] > 0 and outcomes["corrupt"] > 0


def test_page_reader_under_asan(tmp_path):
    """tests/native/fuzz_pages.cpp: 4000 randomly damaged relations through the reader built with
    AddressSanitizer + UBSan (out-of-bounds reads would abort the run)."""
    import os
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_pages")
    src = [os.path.join(root, p) for p in ("tests/native/fuzz_pages.cpp", "vectorchord-bm25_amd/csrc/pages.cpp",
                                           "vectorchord-bm25_amd/csrc/segment.cpp", "vectorchord-bm25_amd/csrc/blake3.cpp", "oracle/oracle.cpp",
                                           "oracle/pages.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-pthread", *src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "fuzz done" in out.stdout
