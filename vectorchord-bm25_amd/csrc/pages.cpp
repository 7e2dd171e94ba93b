// pages.cpp -- host side: a bm25 index relation in the reference's on-disk format -> flattened
// arrays (vbm25_segment) and the growing segment's CSR.  See include/vbm25.h for the contract; the
// reference files restated here are cited there and below.  No device code.
#include "vbm25_internal.h"

#include <cstring>
#include <memory>
#include <unordered_set>

using namespace vbm25;

namespace {

constexpr uint32_t BLCKSZ = 8192, HDR = 24, NONE = 0xffffffffu;

template <class T>
T rd(const uint8_t *p) {
    T v;
    std::memcpy(&v, p, sizeof v);
    return v;
}

struct Corrupt {  // thrown by the page accessors; turned into VBM25_ERR_CORRUPT at the ABI
    const char *what;
    uint32_t page;
};

// One page image: PageHeaderData / ItemIdData / special area, as src/index/storage.rs:49-110 reads them
struct PageView {
    const uint8_t *p;
    uint32_t id;
    uint16_t len() const {
        const uint16_t lower = rd<uint16_t>(p + 12), upper = rd<uint16_t>(p + 14);
        if (lower < HDR || lower > upper || upper > BLCKSZ) throw Corrupt{"page header out of range", id};
        return uint16_t((lower - HDR) / 4);
    }
    // slot i (1-based) -> tuple bytes
    const uint8_t *get(uint16_t i, uint32_t &size) const {
        if (i == 0 || i > len()) throw Corrupt{"slot out of range", id};
        const uint32_t iid = rd<uint32_t>(p + HDR + 4u * (i - 1));
        const uint32_t off = iid & 0x7fff, flags = (iid >> 15) & 3, n = iid >> 17;
        if (flags != 1 /* LP_NORMAL */) throw Corrupt{"line pointer is not LP_NORMAL", id};
        if (off < HDR || off + n > BLCKSZ) throw Corrupt{"line pointer out of range", id};
        size = n;
        return p + off;
    }
    uint32_t next() const {  // Opaque.next, crates/bm25/src/lib.rs:41-46
        const uint16_t special = rd<uint16_t>(p + 16);
        if (special != BLCKSZ - 8) throw Corrupt{"special area is not Opaque", id};
        return rd<uint32_t>(p + special);
    }
};

struct Relation {
    vbm25_read_page_fn fn;
    void *ctx;
    mutable std::unordered_set<uint32_t> walked;  // a page belongs to one tape, once: damaged links must not loop
    PageView read(uint32_t id) const {
        const uint8_t *p = fn(ctx, id);
        if (!p) throw Corrupt{"page cannot be read", id};
        return PageView{p, id};
    }
};

// tape.rs:169-199: every tuple of every page from `first` following Opaque.next
template <class F>
void walk_tape(const Relation &rel, uint32_t first, F &&visit) {
    for (uint32_t cur = first; cur != NONE;) {
        if (!rel.walked.insert(cur).second) throw Corrupt{"page linked twice", cur};
        const PageView pg = rel.read(cur);
        const uint16_t n = pg.len();
        for (uint16_t i = 1; i <= n; ++i) {
            uint32_t size = 0;
            const uint8_t *t = pg.get(i, size);
            visit(cur, i, t, size);
        }
        cur = pg.next();
    }
}

struct Jump {
    uint32_t ptr_vectors, n_docs;
    uint64_t sum_len;
    uint32_t ptr_documents, ptr_tokens, ptr_summaries, ptr_blocks;
};

// search.rs:37-51: Meta (page 0, slot 1) -> k1, b, ptr_jump; Jump (slot 1 of that page)
Jump read_meta_jump(const Relation &rel, double &k1, double &b) {
    uint32_t size = 0;
    const PageView meta_page = rel.read(0);
    const uint8_t *m = meta_page.get(1, size);
    if (size < 72 || std::memcmp(m, "vchordbm", 8) != 0) throw Corrupt{"bad magic number", 0};
    if (rd<uint64_t>(m + 8) != 1) throw Corrupt{"bad version number: REINDEX needed", 0};  // tuples.rs:106-111
    k1 = rd<double>(m + 16);
    b = rd<double>(m + 24);
    const uint32_t ptr_jump = rd<uint32_t>(m + 36);
    const PageView jp = rel.read(ptr_jump);
    const uint8_t *j = jp.get(1, size);
    if (size < 64) throw Corrupt{"jump tuple too short", ptr_jump};
    Jump o;
    o.ptr_vectors = rd<uint32_t>(j + 0);
    o.n_docs = rd<uint32_t>(j + 4);
    o.sum_len = rd<uint64_t>(j + 8);
    o.ptr_documents = rd<uint32_t>(j + 44);
    o.ptr_tokens = rd<uint32_t>(j + 48);
    o.ptr_summaries = rd<uint32_t>(j + 52);
    o.ptr_blocks = rd<uint32_t>(j + 56);
    return o;
}

}  // namespace

struct vbm25_growing {
    std::vector<uint64_t> start{0};
    std::vector<uint8_t> key;
    std::vector<uint32_t> tf;
    std::vector<uint8_t> fieldnorm;
    std::vector<uint16_t> payload;
    std::vector<uint8_t> deleted;
};

extern "C" {

int vbm25_segment_from_pages(vbm25_read_page_fn read_page, void *ctx, vbm25_segment **out) {
    if (!read_page || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
    *out = nullptr;
    try {
        const Relation rel{read_page, ctx, {}};
        auto seg = std::make_unique<vbm25_segment>();
        const Jump jump = read_meta_jump(rel, seg->k1, seg->b);
        seg->n_docs = jump.n_docs;
        seg->sum_len = jump.sum_len;
        // documents (DocumentTuple, tuples.rs:756-781): fieldnorm + payload per document, in id order.
        // `deleted` is not read by search (search.rs:217-229) and is not kept.
        walk_tape(rel, jump.ptr_documents, [&](uint32_t page, uint16_t, const uint8_t *t, uint32_t size) {
            if (size < 8) throw Corrupt{"document tuple too short", page};
            seg->doc_fieldnorm.push_back(t[1]);
            for (int i = 0; i < 3; ++i) seg->doc_payload.push_back(rd<uint16_t>(t + 2 + 2 * i));
        });
        if (seg->doc_fieldnorm.size() != jump.n_docs) throw Corrupt{"document count differs from the Jump tuple", jump.ptr_documents};
        // tokens (TokenTuple, tuples.rs:833-862), ascending key
        struct Tok {
            uint32_t sum_page;
            uint16_t sum_slot;
        };
        std::vector<Tok> toks;
        walk_tape(rel, jump.ptr_tokens, [&](uint32_t page, uint16_t, const uint8_t *t, uint32_t size) {
            if (size < 32) throw Corrupt{"token tuple too short", page};
            seg->term_key.insert(seg->term_key.end(), t, t + 16);
            seg->term_wand_fn.push_back(t[17]);
            toks.push_back({rd<uint32_t>(t + 18), rd<uint16_t>(t + 22)});
            seg->term_df.push_back(rd<uint32_t>(t + 24));
            seg->term_wand_tf.push_back(rd<uint32_t>(t + 28));
        });
        seg->n_terms = uint32_t(toks.size());
        // summaries (SummaryTuple, tuples.rs:900-934): token after token, ceil(df / 128) each (flush.rs:71-125)
        struct Sum {
            uint32_t page;
            uint16_t slot;
            uint32_t blk_page;
            uint16_t blk_slot;
        };
        std::vector<Sum> sums;
        walk_tape(rel, jump.ptr_summaries, [&](uint32_t page, uint16_t slot, const uint8_t *t, uint32_t size) {
            if (size < 24) throw Corrupt{"summary tuple too short", page};
            seg->blk_min_doc.push_back(rd<uint32_t>(t + 0));
            seg->blk_max_doc.push_back(rd<uint32_t>(t + 4));
            sums.push_back({page, slot, rd<uint32_t>(t + 8), rd<uint16_t>(t + 12)});
            seg->blk_n.push_back(t[14]);
            seg->blk_wand_fn.push_back(t[15]);
            seg->blk_wand_tf.push_back(rd<uint32_t>(t + 16));
        });
        seg->n_blocks = uint32_t(sums.size());
        seg->term_first_block.push_back(0);
        uint64_t at = 0;
        for (uint32_t t = 0; t < seg->n_terms; ++t) {
            const uint64_t nb = (uint64_t(seg->term_df[t]) + 127) / 128;
            if (seg->term_df[t] == 0 || at + nb > sums.size()) throw Corrupt{"summaries do not cover the tokens", jump.ptr_summaries};
            if (sums[at].page != toks[t].sum_page || sums[at].slot != toks[t].sum_slot)
                throw Corrupt{"a token's first summary is not where its pointer says", toks[t].sum_page};
            at += nb;
            seg->term_first_block.push_back(uint32_t(at));
        }
        if (at != sums.size()) throw Corrupt{"summaries left over after the last token", jump.ptr_summaries};
        // blocks (BlockTuple, tuples.rs:973-1025), same order as the summaries; bodies copied as they are
        uint32_t j = 0;
        seg->blk_off8.push_back(0);
        walk_tape(rel, jump.ptr_blocks, [&](uint32_t page, uint16_t slot, const uint8_t *t, uint32_t size) {
            if (j >= sums.size()) throw Corrupt{"more blocks than summaries", page};
            if (sums[j].blk_page != page || sums[j].blk_slot != slot) throw Corrupt{"a summary's block is not where its pointer says", page};
            if (size < 16) throw Corrupt{"block tuple too short", page};
            const uint8_t md = t[0], mt = t[1];
            const uint16_t ds = rd<uint16_t>(t + 2), de = rd<uint16_t>(t + 4), ts = rd<uint16_t>(t + 6), te = rd<uint16_t>(t + 8);
            const uint32_t n = seg->blk_n[j];
            const uint32_t ld = (md >> 7) ? (md & 127u) * n : 16u * (md & 127u);
            const uint32_t lt = (mt >> 7) ? (mt & 127u) * n : 16u * (mt & 127u);
            if (ds != 16 || uint32_t(de - ds) != ld || ts != ((de + 7u) & ~7u) || uint32_t(te - ts) != lt || te > size)
                throw Corrupt{"block tuple ranges do not match its codec metadata", page};
            // flattened body = doc-id bytes, pad to 8, tf bytes, pad to 8 (zero padding)
            seg->blob.insert(seg->blob.end(), t + ds, t + de);
            seg->blob.resize((seg->blob.size() + 7) & ~size_t(7), 0);
            seg->blob.insert(seg->blob.end(), t + ts, t + te);
            seg->blob.resize((seg->blob.size() + 7) & ~size_t(7), 0);
            seg->blk_meta_doc.push_back(md);
            seg->blk_meta_tf.push_back(mt);
            seg->blk_off8.push_back(uint32_t(seg->blob.size() / 8));
            ++j;
        });
        if (j != sums.size()) throw Corrupt{"fewer blocks than summaries", jump.ptr_blocks};
        {   // the same structural checks vbm25_index_create makes: a segment handed out is valid or refused
            vbm25_index_desc d;
            seg->desc(&d);
            if (int rc = check_desc(&d)) return rc;
        }
        *out = seg.release();
        return VBM25_OK;
    } catch (const Corrupt &c) {
        return set_error(VBM25_ERR_CORRUPT, "data corruption: %s (page %u)", c.what, c.page);
    } catch (const std::bad_alloc &) {
        return set_error(VBM25_ERR_NOMEM, "out of host memory while flattening the index");
    } catch (const std::exception &e) {
        return set_error(VBM25_ERR_INVALID, "internal error: %s", e.what());
    }
}

int vbm25_growing_from_pages(vbm25_read_page_fn read_page, void *ctx, vbm25_growing **out) {
    if (!read_page || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
    *out = nullptr;
    try {
        const Relation rel{read_page, ctx, {}};
        double k1, b;
        const Jump jump = read_meta_jump(rel, k1, b);
        if (jump.ptr_vectors == NONE) throw Corrupt{"no vectors tape", 0};  // search.rs:85
        auto g = std::make_unique<vbm25_growing>();
        bool open = false;  // a _2 tuple started a document (search.rs:87,94-96)
        auto elements = [&](uint32_t page, const uint8_t *t, uint32_t size, uint32_t hdr_off) {
            const uint16_t s = rd<uint16_t>(t + hdr_off), e = rd<uint16_t>(t + hdr_off + 2);
            if (s > e || e > size || (e - s) % 20) throw Corrupt{"vector tuple element range", page};
            for (uint32_t p = s; p < e; p += 20) {
                g->key.insert(g->key.end(), t + p, t + p + 16);
                g->tf.push_back(rd<uint32_t>(t + p + 16));
            }
        };
        walk_tape(rel, jump.ptr_vectors, [&](uint32_t page, uint16_t, const uint8_t *t, uint32_t size) {
            if (size < 16) throw Corrupt{"vector tuple too short", page};
            switch (rd<uint64_t>(t)) {
            case 2:  // fieldnorm: starts a document
                if (open) {  // a _2 while a document is open: the reference overwrites its state (search.rs:94-96),
                             // i.e. drops the unfinished document (an insert that failed before its _0)
                    g->fieldnorm.pop_back();
                    g->key.resize(16 * g->start.back());
                    g->tf.resize(g->start.back());
                }
                g->fieldnorm.push_back(t[8]);
                open = true;
                break;
            case 1:  // continuation
                if (!open) throw Corrupt{"vector continuation without a start", page};
                elements(page, t, size, 8);
                break;
            case 0:  // last tuple of the document
                if (!open || size < 24) throw Corrupt{"vector end without a start", page};
                elements(page, t, size, 16);
                g->deleted.push_back(t[8]);
                for (int i = 0; i < 3; ++i) g->payload.push_back(rd<uint16_t>(t + 10 + 2 * i));
                g->start.push_back(g->tf.size());
                open = false;
                break;
            default:
                throw Corrupt{"vector tuple tag", page};
            }
        });
        if (open) {  // a _2 without its _0: the insert had not finished when the pages were read
            g->fieldnorm.pop_back();
            g->key.resize(16 * g->start.back());
            g->tf.resize(g->start.back());
        }
        *out = g.release();
        return VBM25_OK;
    } catch (const Corrupt &c) {
        return set_error(VBM25_ERR_CORRUPT, "data corruption: %s (page %u)", c.what, c.page);
    } catch (const std::bad_alloc &) {
        return set_error(VBM25_ERR_NOMEM, "out of host memory while reading the growing segment");
    } catch (const std::exception &e) {
        return set_error(VBM25_ERR_INVALID, "internal error: %s", e.what());
    }
}

// Meta and Jump tuple bytes of the relation (the two tuples that change when the sealed segment is replaced)
static int meta_jump_bytes(vbm25_read_page_fn read_page, void *ctx, uint8_t *meta72, uint8_t *jump64) {
    const Relation rel{read_page, ctx, {}};
    uint32_t size = 0;
    const PageView meta_page = rel.read(0);
    const uint8_t *m = meta_page.get(1, size);
    if (size < 72 || std::memcmp(m, "vchordbm", 8) != 0) throw Corrupt{"bad magic number", 0};
    if (rd<uint64_t>(m + 8) != 1) throw Corrupt{"bad version number: REINDEX needed", 0};
    std::memcpy(meta72, m, 72);
    if (jump64) {
        const uint32_t ptr_jump = rd<uint32_t>(m + 36);
        const PageView jp = rel.read(ptr_jump);
        const uint8_t *j = jp.get(1, size);
        if (size < 64) throw Corrupt{"jump tuple too short", ptr_jump};
        std::memcpy(jump64, j, 64);
    }
    return VBM25_OK;
}

int vbm25_pages_seed(vbm25_read_page_fn read_page, void *ctx, uint8_t *seed32) {
    if (!read_page || !seed32) return set_error(VBM25_ERR_INVALID, "NULL argument");
    try {
        uint8_t meta[72];
        meta_jump_bytes(read_page, ctx, meta, nullptr);
        std::memcpy(seed32, meta + 40, 32);  // MetaTuple.seed, tuples.rs:48-57
        return VBM25_OK;
    } catch (const Corrupt &c) {
        return set_error(VBM25_ERR_CORRUPT, "data corruption: %s (page %u)", c.what, c.page);
    } catch (...) {
        return set_error(VBM25_ERR_INVALID, "internal error");
    }
}

int vbm25_pages_fingerprint(vbm25_read_page_fn read_page, void *ctx, uint8_t *out32) {
    if (!read_page || !out32) return set_error(VBM25_ERR_INVALID, "NULL argument");
    try {
        uint8_t buf[72 + 64];
        meta_jump_bytes(read_page, ctx, buf, buf + 72);
        blake3(nullptr, buf, sizeof buf, out32, 32);
        return VBM25_OK;
    } catch (const Corrupt &c) {
        return set_error(VBM25_ERR_CORRUPT, "data corruption: %s (page %u)", c.what, c.page);
    } catch (...) {
        return set_error(VBM25_ERR_INVALID, "internal error");
    }
}

int vbm25_growing_get_desc(const vbm25_growing *g, vbm25_growing_desc *out) {
    if (!g || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
    out->n_docs = uint32_t(g->start.size() - 1);
    out->_pad = 0;
    out->n_elements = g->tf.size();
    out->start = g->start.data();
    out->key = g->key.data();
    out->tf = g->tf.data();
    out->fieldnorm = g->fieldnorm.data();
    out->payload = g->payload.data();
    out->deleted = g->deleted.data();
    return VBM25_OK;
}

void vbm25_growing_free(vbm25_growing *g) { delete g; }

}  // extern "C"
