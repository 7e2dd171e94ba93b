#!/usr/bin/env python3
"""A small bm25 index relation assembled with nothing but struct.pack, at the offsets of the reference's
tuple definitions (crates/bm25/src/tuples.rs:48-1102, crates/bm25/src/lib.rs:41-46 for the page trailer,
src/index/storage.rs:49-170 for the PostgreSQL page primitives, compression.rs:36-136 + crates/simd for
the block payloads, flush.rs:40-158 for what goes where).  It does NOT use oracle/pages.cpp or the library:
it is a third, independent derivation of the format, so that the product's page reader
(vbm25_segment_from_pages / vbm25_growing_from_pages) is not only tested against a writer by the same
author's other hand.  Writes page_fixture.bin (8192-byte pages) and page_fixture.json (what a reader must
return).  Run from anywhere:  python tests/golden/make_page_fixture.py
"""
import json
import math
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
BLCKSZ, NONE = 8192, 0xFFFFFFFF
K1, B = 1.2, 0.75
SEED = bytes((7 * i + 3) & 0xFF for i in range(32))


def page(tuples, nxt=NONE, flags=0):
    """PageHeaderData (24 B) | ItemIdData x n | free | tuples (downwards, MAXALIGNed) | Opaque{next, flags}."""
    buf = bytearray(BLCKSZ)
    special = BLCKSZ - 8
    upper = special
    lps = []
    for t in tuples:
        t = t + b"\0" * (-len(t) % 8)  # every tuple is padded to ALIGN = 8 when serialised (tuples.rs:20)
        size = len(t)
        upper -= (size + 7) & ~7
        buf[upper:upper + size] = t
        lps.append(upper | (1 << 15) | (size << 17))  # lp_off:15, lp_flags:2 = LP_NORMAL, lp_len:15
    lower = 24 + 4 * len(tuples)
    assert lower <= upper
    struct.pack_into("<QHHHHHHI", buf, 0, 0, 0, 0, lower, upper, special, BLCKSZ | 4, 0)
    for i, lp in enumerate(lps):
        struct.pack_into("<I", buf, 24 + 4 * i, lp)
    struct.pack_into("<II", buf, special, nxt, flags)
    return bytes(buf)


def key(s):
    assert len(s) < 16 and b"\0" not in s
    return s + b"\0" * (16 - len(s))


def bitpack(values, width):
    """crates/simd/src/bitpacking.rs:58-98: value i -> lane i % 4, step i // 4; lane l is an LSB-first stream of
    width-bit fields whose 32-bit word w lives at byte 16 w + 4 l."""
    assert len(values) == 128
    out = bytearray(16 * width)
    for i, v in enumerate(values):
        assert v < (1 << width)
        lane, step = i % 4, i // 4
        for k in range(width):
            if (v >> k) & 1:
                pos = step * width + k
                w, bit = pos >> 5, pos & 31
                out[16 * w + 4 * lane + (bit >> 3)] |= 1 << (bit & 7)
    return bytes(out)


def bytepack(values, width):
    return b"".join(int(v).to_bytes(width, "little") for v in values)


def width_bits(values):
    o = 0
    for v in values:
        o |= v
    return o.bit_length()


def block_payload(docs, tfs, min_doc):
    """compression.rs:36-63: full blocks bit-packed (doc ids as d1 deltas from min_doc), tails byte-packed."""
    deltas = [docs[0] - min_doc] + [docs[i] - docs[i - 1] for i in range(1, len(docs))]
    if len(docs) == 128:
        bd, bt = width_bits(deltas), width_bits(tfs)
        return bd, bitpack(deltas, bd), bt, bitpack(tfs, bt)
    wd = max(1, (width_bits(deltas) + 7) // 8)
    wt = max(1, (width_bits(tfs) + 7) // 8)
    return 0x80 | wd, bytepack(deltas, wd), 0x80 | wt, bytepack(tfs, wt)


def block_tuple(md, dbytes, mt, tbytes):
    ds = 16
    de = ds + len(dbytes)
    ts = (de + 7) & ~7
    te = ts + len(tbytes)
    body = bytearray((te + 7) & ~7)
    struct.pack_into("<BBHHHH", body, 0, md, mt, ds, de, ts, te)
    body[ds:de] = dbytes
    body[ts:te] = tbytes
    return bytes(body)


def main():
    n_docs = 200
    lens = [5 + d % 30 for d in range(n_docs)]          # <= 40: the fieldnorm code is the length itself (bm25.rs:15-60)
    payload = [(0, d, 1 + d % 7) for d in range(n_docs)]
    terms = [
        (key(b"alpha"), list(range(150)), [1 + d % 3 for d in range(150)]),
        (key(b"beta"), list(range(7, 200, 13)), [2] * len(range(7, 200, 13))),
    ]
    avgdl = sum(lens) / n_docs

    def tf_score(fn, tf):  # bm25.rs:291-295
        return (tf * (K1 + 1.0)) / (tf + K1 * (1.0 - B + B * lens_of_fn(fn) / avgdl))

    def lens_of_fn(fn):
        return float(fn)

    def wand(docs, tfs):  # first maximiser, strict < (bm25.rs:297-332)
        best, pair = -1.0, (0, 0)
        for d, tf in zip(docs, tfs):
            s = tf_score(lens[d], tf)
            if best < s:
                best, pair = s, (lens[d], tf)
        return best, pair

    P_META, P_JUMP, P_DOCS, P_TOKENS, P_SUMS, P_BLOCKS, P_VEC0, P_VEC1, P_ADOC, P_ATOK = range(10)
    # ---- blocks / summaries / tokens (flush.rs:71-125: per token its summaries, in order; blocks likewise)
    blocks, sums, toks, exp_blk = [], [], [], []
    for tkey, docs, tfs in terms:
        first_sum_slot = len(sums) + 1
        tbest, tpair = -1.0, (0, 0)
        for s in range(0, len(docs), 128):
            bd, bt = docs[s:s + 128], tfs[s:s + 128]
            md, dbytes, mt, tbytes = block_payload(bd, bt, bd[0])
            blocks.append(block_tuple(md, dbytes, mt, tbytes))
            score, pair = wand(bd, bt)
            if tbest < score:
                tbest, tpair = score, pair
            sums.append(struct.pack("<IIIHBBI4x", bd[0], bd[-1], P_BLOCKS, len(blocks), len(bd), pair[0], pair[1]))
            exp_blk.append(dict(min_doc=bd[0], max_doc=bd[-1], n=len(bd), wand_fn=pair[0], wand_tf=pair[1], meta_doc=md,
                                meta_tf=mt, doc_bytes=dbytes.hex(), tf_bytes=tbytes.hex()))
        toks.append(tkey + struct.pack("<BBIHII", 0, tpair[0], P_SUMS, first_sum_slot, len(docs), tpair[1]))
    docs_t = [struct.pack("<BBHHH", 0, lens[d], *payload[d]) for d in range(n_docs)]
    # ---- growing tape (tuples.rs:326-426): _2 starts a document, _1 continues it, _0 ends it
    def elems(lst):
        return b"".join(key(k) + struct.pack("<I", v) for k, v in lst)

    def v2(fn):
        return struct.pack("<QB7x", 2, fn)

    def v1(lst):
        e = elems(lst)
        return struct.pack("<QHH4x", 1, 16, 16 + len(e)) + e

    def v0(deleted, pl, lst):
        e = elems(lst)
        return struct.pack("<QBxHHHHH4x", 0, deleted, pl[0], pl[1], pl[2], 24, 24 + len(e)) + e

    vec0 = [v2(9), v0(0, (9, 9, 9), [(b"alpha", 2), (b"gamma", 1)]),
            v2(4), v1([(b"beta", 7)])]                                   # an insert that failed before its _0
    vec1 = [v2(12), v1([(b"alpha", 1)]), v0(1, (1, 2, 3), [(b"beta", 3), (b"zeta", 5)]),
            v2(3)]                                                       # an insert still in progress
    meta = b"vchordbm" + struct.pack("<QddII", 1, K1, B, NONE, P_JUMP) + SEED
    assert len(meta) == 72
    jump = struct.pack("<IIQHHIIIIIIIIII4x", P_VEC0, n_docs, sum(lens), 2036, 680, 0, P_ADOC, NONE, 0, P_ATOK, NONE,
                       P_DOCS, P_TOKENS, P_SUMS, P_BLOCKS)
    assert len(jump) == 64
    adoc = struct.pack("<HH4xI", 8, 12, P_DOCS)                          # AddressDocumentsTuple: one child page
    atok = struct.pack("<HH4x", 8, 28) + terms[-1][0] + struct.pack("<I", P_TOKENS)  # one Edge{last key, page}
    pages = [page([meta]), page([jump]), page(docs_t), page(toks), page(sums), page(blocks),
             page(vec0, nxt=P_VEC1), page(vec1), page([adoc]), page([atok])]
    open(os.path.join(HERE, "page_fixture.bin"), "wb").write(b"".join(pages))
    first_block, at = [0], 0
    for _, docs, _ in terms:
        at += (len(docs) + 127) // 128
        first_block.append(at)
    exp = dict(
        n_docs=n_docs, sum_len=sum(lens), k1=K1, b=B, seed=SEED.hex(),
        term_key=[t[0].hex() for t in terms], term_df=[len(t[1]) for t in terms],
        term_wand=[list(struct.unpack_from("<B", tk, 17)) + list(struct.unpack_from("<I", tk, 28)) for tk in toks],
        term_first_block=first_block, blocks=exp_blk, doc_fieldnorm=lens, doc_payload=payload,
        postings=[dict(docs=t[1], tfs=t[2]) for t in terms],
        growing=dict(fieldnorm=[9, 12], deleted=[0, 1], payload=[[9, 9, 9], [1, 2, 3]],
                     docs=[[[key(b"alpha").hex(), 2], [key(b"gamma").hex(), 1]],
                           [[key(b"alpha").hex(), 1], [key(b"beta").hex(), 3], [key(b"zeta").hex(), 5]]]))
    json.dump(exp, open(os.path.join(HERE, "page_fixture.json"), "w"), indent=1)
    print(f"{len(pages)} pages, {len(blocks)} blocks written")


if __name__ == "__main__":
    main()
