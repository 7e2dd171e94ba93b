// TEST INFRASTRUCTURE ONLY (see oracle.h).  Restatement of how the reference lays a bm25 index out
// in PostgreSQL pages, so that the product's page reader (include/vbm25.h: vbm25_segment_from_pages,
// vbm25_growing_from_pages) can be tested against relations in the reference's on-disk format.
//
// Follows, citing /root/reference:
//   page primitives   src/index/storage.rs:49-170 over PostgreSQL's PageInit / PageAddItemExtended /
//                     PageGetFreeSpace (bufpage.c: 24-byte header, 4-byte line pointers, MAXALIGN 8,
//                     special area = crate::Opaque{next, flags}, crates/bm25/src/lib.rs:41-46)
//   tapes             crates/bm25/src/tape.rs:21-167
//   tuples            crates/bm25/src/tuples.rs (Meta 48-94, Jump 141-203, Vector 326-426, AddressDocuments
//                     602-650, AddressTokens 679-728, Document 756-781, Token 833-862, Summary 900-934,
//                     Block 973-1025, Pointer 1070-1086, Edge 1088-1102)
//   build             crates/bm25/src/build.rs:22-71, flush.rs:40-158 (tape order and page allocation order),
//                     address_documents.rs:26-73, address_tokens.rs:26-60
//   insert            crates/bm25/src/insert.rs:23-79
// PARITY UNPINNED: no page image of the reference is available here (no PostgreSQL, no Rust toolchain);
// what pins this file are the per-page tuple counts the reference's format implies (226 tokens, 291
// summaries, 680 documents per page; 2036 / 407 entries per address page) checked in
// tests/test_pages.py.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "oracle.h"

namespace {

constexpr uint32_t BLCKSZ = 8192, HDR = 24, SPECIAL = 8, NONE = 0xffffffffu;

struct Page {
    uint8_t b[BLCKSZ];
    uint16_t &u16(size_t off) { return *reinterpret_cast<uint16_t *>(b + off); }
    uint32_t &u32(size_t off) { return *reinterpret_cast<uint32_t *>(b + off); }
    uint16_t lower() { return u16(12); }
    uint16_t upper() { return u16(14); }
    void init(uint32_t next) {  // PageInit(page, BLCKSZ, sizeof(Opaque)) + opaque
        std::memset(b, 0, BLCKSZ);
        u16(12) = HDR;                    // pd_lower
        u16(14) = BLCKSZ - SPECIAL;       // pd_upper
        u16(16) = BLCKSZ - SPECIAL;       // pd_special
        u16(18) = BLCKSZ | 4;             // pd_pagesize_version (PG_PAGE_LAYOUT_VERSION 4)
        u32(BLCKSZ - SPECIAL) = next;     // Opaque.next
        u32(BLCKSZ - SPECIAL + 4) = 0;    // Opaque.flags
    }
    uint16_t len() { return uint16_t((lower() - HDR) / 4); }
    uint16_t freespace() {  // PageGetFreeSpace: room for one more line pointer is taken off
        const int space = int(upper()) - int(lower());
        return space < 4 ? 0 : uint16_t(space - 4);
    }
    uint16_t alloc(const std::vector<uint8_t> &t) {  // PageAddItemExtended at the end, flags 0
        const uint32_t aligned = (uint32_t(t.size()) + 7u) & ~7u;
        const int lo = lower() + 4, up = int(upper()) - int(aligned);
        if (lo > up) return 0;
        const uint16_t n = len();
        u32(HDR + 4 * n) = uint32_t(up) | (1u << 15) /* LP_NORMAL */ | (uint32_t(t.size()) << 17);
        std::memcpy(b + up, t.data(), t.size());
        u16(12) = uint16_t(lo);
        u16(14) = uint16_t(up);
        return uint16_t(n + 1);
    }
    uint32_t &next() { return u32(BLCKSZ - SPECIAL); }
};

}  // namespace

struct orc_pages {
    std::vector<Page> pages;
    uint32_t alloc(uint32_t next) {
        pages.emplace_back();
        pages.back().init(next);
        return uint32_t(pages.size() - 1);
    }
};

namespace {

using Bytes = std::vector<uint8_t>;
template <class T>
void put(Bytes &b, size_t off, T v) {
    std::memcpy(b.data() + off, &v, sizeof v);
}
void pad8(Bytes &b) {
    while (b.size() % 8) b.push_back(0);
}

struct Tape {  // tape.rs:21-108
    orc_pages *rel;
    uint32_t head, first;
    static Tape create(orc_pages *r) {
        const uint32_t p = r->alloc(NONE);
        return Tape{r, p, p};
    }
    void move() {
        if (rel->pages[head].len() == 0) throw std::logic_error("a clear page cannot accommodate a single tuple");
        const uint32_t n = rel->alloc(NONE);
        rel->pages[head].next() = n;
        head = n;
    }
    std::pair<uint32_t, uint16_t> push(const Bytes &t) {
        if (uint16_t i = rel->pages[head].alloc(t)) return {head, i};
        const uint32_t n = rel->alloc(NONE);
        rel->pages[head].next() = n;
        head = n;
        if (uint16_t i = rel->pages[head].alloc(t)) return {head, i};
        throw std::logic_error("a free page cannot accommodate a single tuple");
    }
    std::pair<uint32_t, uint16_t> tape_put(const Bytes &t) {
        if (uint16_t i = rel->pages[head].alloc(t)) return {head, i};
        throw std::logic_error("a free page cannot accommodate a single tuple");
    }
    uint16_t freespace() { return rel->pages[head].freespace(); }
};

struct BackTape {  // tape.rs:110-167
    orc_pages *rel;
    uint32_t head;
    static BackTape create(orc_pages *r) { return BackTape{r, r->alloc(NONE)}; }
    uint16_t freespace() { return rel->pages[head].freespace(); }
    uint32_t tape_put(const Bytes &t) {
        if (rel->pages[head].alloc(t)) return head;
        throw std::logic_error("a free page cannot accommodate a single tuple");
    }
    void move() {
        if (rel->pages[head].len() == 0) throw std::logic_error("a clear page cannot accommodate a single tuple");
        head = rel->alloc(head);
    }
};

size_t fit(uint16_t freespace, size_t elem) {  // AddressDocumentsTuple::fit / AddressTokensTuple::fit
    long fs = freespace;
    fs &= ~7L;
    fs -= 8;
    fs &= ~7L;
    if (fs < 0) throw std::logic_error("a blank page cannot fit a single tuple");
    return size_t(fs) / elem;
}

Bytes ranged(const uint8_t *data, size_t bytes) {  // 8-byte header {s, e, pad4} + payload, padded
    Bytes t(8, 0);
    const uint16_t s = uint16_t(t.size());
    t.insert(t.end(), data, data + bytes);
    const uint16_t e = uint16_t(t.size());
    pad8(t);
    put(t, 0, s);
    put(t, 2, e);
    return t;
}

}  // namespace

extern "C" {

orc_pages *orc_pages_build(const orc_index *ix, const uint8_t *seed32) {
    orc_index_view v;
    orc_index_get_view(ix, &v);
    auto rel = new orc_pages;
    // build.rs:34-36: the meta tape comes first (page 0); its tuple is written last
    Tape meta = Tape::create(rel);
    if (meta.first != 0) throw std::logic_error("meta page must be page 0");

    // ---- flush.rs:50-66: documents
    std::vector<std::pair<uint32_t, uint16_t>> map_documents;
    Tape tape_documents = Tape::create(rel);
    for (uint32_t d = 0; d < v.n_docs; ++d) {
        Bytes t(8, 0);  // DocumentTupleHeader {deleted, fieldnorm, payload[3]}
        t[1] = v.doc_fieldnorm[d];
        std::memcpy(t.data() + 2, v.doc_payload + 3ull * d, 6);
        map_documents.push_back(tape_documents.push(t));
    }
    // ---- flush.rs:70-139: tokens / summaries / blocks
    Tape tape_tokens = Tape::create(rel), tape_summaries = Tape::create(rel), tape_blocks = Tape::create(rel);
    struct TokMap {
        const uint8_t *key;
        std::pair<uint32_t, uint16_t> at;
    };
    std::vector<TokMap> map_tokens;
    for (uint32_t t = 0; t < v.n_terms; ++t) {
        std::pair<uint32_t, uint16_t> wptr_summaries{tape_summaries.first, 1};
        for (uint32_t j = v.term_first_block[t]; j < v.term_first_block[t + 1]; ++j) {
            const uint8_t *body = v.blob + 8ull * v.blk_off8[j];
            const uint8_t md = v.blk_meta_doc[j], mt = v.blk_meta_tf[j];
            const uint32_t n = v.blk_n[j];
            const uint32_t ld = (md >> 7) ? (md & 127u) * n : 16u * (md & 127u);
            const uint32_t lt = (mt >> 7) ? (mt & 127u) * n : 16u * (mt & 127u);
            Bytes bt(16, 0);  // BlockTupleHeader
            const uint16_t ds = uint16_t(bt.size());
            bt.insert(bt.end(), body, body + ld);
            const uint16_t de = uint16_t(bt.size());
            pad8(bt);
            const uint16_t ts = uint16_t(bt.size());
            const uint8_t *tb = body + ((ld + 7u) & ~7u);
            bt.insert(bt.end(), tb, tb + lt);
            const uint16_t te = uint16_t(bt.size());
            pad8(bt);
            bt[0] = md;
            bt[1] = mt;
            put(bt, 2, ds);
            put(bt, 4, de);
            put(bt, 6, ts);
            put(bt, 8, te);
            const auto wptr_block = tape_blocks.push(bt);
            Bytes st(24, 0);  // SummaryTupleHeader
            put(st, 0, v.blk_min_doc[j]);
            put(st, 4, v.blk_max_doc[j]);
            put(st, 8, wptr_block.first);    // Pointer packed(2): x u32, y u16
            put(st, 12, wptr_block.second);
            st[14] = uint8_t(n);
            st[15] = v.blk_wand_fn[j];
            put(st, 16, v.blk_wand_tf[j]);
            const auto wptr_summary = tape_summaries.push(st);
            if (j == v.term_first_block[t]) wptr_summaries = wptr_summary;
        }
        Bytes tt(32, 0);  // TokenTupleHeader
        std::memcpy(tt.data(), v.term_key + 16ull * t, 16);
        tt[17] = v.term_wand_fn[t];
        put(tt, 18, wptr_summaries.first);
        put(tt, 22, wptr_summaries.second);
        put(tt, 24, v.term_df[t]);
        put(tt, 28, v.term_wand_tf[t]);
        map_tokens.push_back({v.term_key + 16ull * t, tape_tokens.push(tt)});
    }
    // ---- address_documents.rs:26-73
    uint16_t width_1_documents, width_0_documents;
    uint32_t depth_documents = 0, start_documents, free_documents;
    {
        BackTape tape = BackTape::create(rel);
        width_1_documents = uint16_t(fit(tape.freespace(), 4));
        uint16_t w0 = 1;
        if (!map_documents.empty()) {
            w0 = 0;
            while (w0 < map_documents.size() && map_documents[w0].first == map_documents[0].first) ++w0;
        }
        width_0_documents = w0;
        std::vector<uint32_t> buffer;
        for (size_t i = 0; i < map_documents.size(); i += w0) buffer.push_back(map_documents[i].first);
        while (buffer.size() > 1) {
            ++depth_documents;
            std::vector<uint32_t> cur;
            cur.swap(buffer);
            for (size_t i = 0; i < cur.size(); i += width_1_documents) {
                const size_t n = std::min<size_t>(width_1_documents, cur.size() - i);
                buffer.push_back(tape.tape_put(ranged(reinterpret_cast<const uint8_t *>(cur.data() + i), 4 * n)));
                tape.move();
            }
        }
        start_documents = buffer.empty() ? NONE : buffer[0];
        free_documents = tape.head;
    }
    // ---- address_tokens.rs:26-60
    uint32_t depth_tokens = 0, start_tokens, free_tokens;
    {
        BackTape tape = BackTape::create(rel);
        const size_t width_1 = fit(tape.freespace(), 20);
        struct Edge {
            uint8_t key[16];
            uint32_t value;
        };
        static_assert(sizeof(Edge) == 20, "Edge is repr(C): 16 + 4");
        std::vector<Edge> buffer;
        for (size_t i = 0; i < map_tokens.size();) {
            size_t j = i;
            while (j + 1 < map_tokens.size() && map_tokens[j + 1].at.first == map_tokens[i].at.first) ++j;
            Edge e;
            std::memcpy(e.key, map_tokens[j].key, 16);
            e.value = map_tokens[j].at.first;
            buffer.push_back(e);
            i = j + 1;
        }
        while (buffer.size() > 1) {
            ++depth_tokens;
            std::vector<Edge> cur;
            cur.swap(buffer);
            for (size_t i = 0; i < cur.size(); i += width_1) {
                const size_t n = std::min(width_1, cur.size() - i);
                Edge e = cur[i + n - 1];
                e.value = tape.tape_put(ranged(reinterpret_cast<const uint8_t *>(cur.data() + i), 20 * n));
                buffer.push_back(e);
                tape.move();
            }
        }
        start_tokens = buffer.empty() ? NONE : buffer[0].value;
        free_tokens = tape.head;
    }
    // ---- build.rs:40-70
    Tape tape_vectors = Tape::create(rel);
    Tape tape_jump = Tape::create(rel);
    Bytes jt(64, 0);  // JumpTupleHeader
    put(jt, 0, tape_vectors.first);
    put(jt, 4, v.n_docs);
    put(jt, 8, v.sum_len);
    put(jt, 16, width_1_documents);
    put(jt, 18, width_0_documents);
    put(jt, 20, depth_documents);
    put(jt, 24, start_documents);
    put(jt, 28, free_documents);
    put(jt, 32, depth_tokens);
    put(jt, 36, start_tokens);
    put(jt, 40, free_tokens);
    put(jt, 44, tape_documents.first);
    put(jt, 48, tape_tokens.first);
    put(jt, 52, tape_summaries.first);
    put(jt, 56, tape_blocks.first);
    const auto ptr_jump = tape_jump.push(jt);
    if (ptr_jump.second != 1) throw std::logic_error("jump tuple must be slot 1");
    Tape tape_lock = Tape::create(rel);
    Bytes mt(72, 0);  // tag + MetaTupleHeader
    std::memcpy(mt.data(), "vchordbm", 8);
    put(mt, 8, uint64_t(1));
    put(mt, 16, v.k1);
    put(mt, 24, v.b);
    put(mt, 32, tape_lock.first);
    put(mt, 36, ptr_jump.first);
    if (seed32) std::memcpy(mt.data() + 40, seed32, 32);
    meta.push(mt);
    return rel;
}

// insert.rs:23-79.  Elements = (key[16], tf) in the document's order.
void orc_pages_insert(orc_pages *rel, const uint16_t *payload3, uint32_t n_elem, const uint8_t *keys,
                      const uint32_t *tfs) {
    uint64_t length = 0;  // vector.rs:77-83: saturating sum of the values
    for (uint32_t i = 0; i < n_elem; ++i) length = std::min<uint64_t>(length + tfs[i], 0xffffffffull);
    const uint8_t fieldnorm = orc_length_to_fieldnorm(uint32_t(length));
    Page &p0 = rel->pages[0];
    const uint32_t ptr_jump = *reinterpret_cast<uint32_t *>(p0.b + (p0.u32(HDR) & 0x7fff) + 36);
    Page &pj = rel->pages[ptr_jump];
    uint32_t current = *reinterpret_cast<uint32_t *>(pj.b + (pj.u32(HDR) & 0x7fff));  // ptr_vectors
    while (rel->pages[current].next() != NONE) current = rel->pages[current].next();
    Tape tape{rel, current, current};
    Bytes t2(16, 0);  // tag 2 + VectorTupleHeader2
    put(t2, 0, uint64_t(2));
    t2[8] = fieldnorm;
    tape.push(t2);
    std::vector<uint8_t> elems(20ull * n_elem);
    for (uint32_t i = 0; i < n_elem; ++i) {
        std::memcpy(elems.data() + 20ull * i, keys + 16ull * i, 16);
        std::memcpy(elems.data() + 20ull * i + 16, tfs + i, 4);
    }
    size_t done = 0;
    for (;;) {
        const size_t remain = n_elem - done;
        const uint16_t freespace = tape.freespace();
        const size_t size0 = 8 + 16 + ((remain * 20 + 7) & ~size_t(7));  // VectorTuple::estimate_size_0
        if (size0 <= freespace) {
            Bytes t(24, 0);  // tag 0 + VectorTupleHeader0 {deleted, pad, payload, elements_s, elements_e, pad4}
            const uint16_t s = uint16_t(t.size());
            t.insert(t.end(), elems.begin() + 20 * done, elems.end());
            const uint16_t e = uint16_t(t.size());
            pad8(t);
            std::memcpy(t.data() + 10, payload3, 6);
            put(t, 16, s);
            put(t, 18, e);
            tape.tape_put(t);
            break;
        }
        long fs = freespace;  // VectorTuple::fit_1
        fs &= ~7L;
        fs -= 8;
        fs &= ~7L;
        fs -= 8;
        fs &= ~7L;
        if (fs >= 0) {
            const size_t w = std::min<size_t>(size_t(fs) / 20, remain);
            Bytes t(16, 0);  // tag 1 + VectorTupleHeader1 {elements_s, elements_e, pad4}
            put(t, 0, uint64_t(1));
            const uint16_t s = uint16_t(t.size());
            t.insert(t.end(), elems.begin() + 20 * done, elems.begin() + 20 * (done + w));
            const uint16_t e = uint16_t(t.size());
            pad8(t);
            put(t, 8, s);
            put(t, 10, e);
            tape.tape_put(t);
            done += w;
        } else {
            tape.move();
        }
    }
}

/* bulkdelete.rs marks documents deleted in place; here: the flag of the i-th sealed document, and
 * the flag of the n-th (0-based) unsealed document's final VectorTuple. */
void orc_pages_mark_deleted_growing(orc_pages *rel, uint32_t nth) {
    Page &p0 = rel->pages[0];
    const uint32_t ptr_jump = *reinterpret_cast<uint32_t *>(p0.b + (p0.u32(HDR) & 0x7fff) + 36);
    Page &pj = rel->pages[ptr_jump];
    uint32_t current = *reinterpret_cast<uint32_t *>(pj.b + (pj.u32(HDR) & 0x7fff));
    uint32_t seen = 0;
    while (current != NONE) {
        Page &p = rel->pages[current];
        for (uint16_t i = 0; i < p.len(); ++i) {
            uint8_t *t = p.b + (p.u32(HDR + 4 * i) & 0x7fff);
            uint64_t tag;
            std::memcpy(&tag, t, 8);
            if (tag == 0 && seen++ == nth) {
                t[8] = 1;  // VectorTupleHeader0.deleted
                return;
            }
        }
        current = p.next();
    }
}

uint32_t orc_pages_count(const orc_pages *rel) { return uint32_t(rel->pages.size()); }
const uint8_t *orc_pages_get(const orc_pages *rel, uint32_t page_id) {
    return page_id < rel->pages.size() ? rel->pages[page_id].b : nullptr;
}
uint8_t *orc_pages_get_mut(orc_pages *rel, uint32_t page_id) {
    return page_id < rel->pages.size() ? rel->pages[page_id].b : nullptr;
}
void orc_pages_free(orc_pages *rel) { delete rel; }

}  // extern "C"
