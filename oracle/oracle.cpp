// oracle.cpp -- CPU restatement of VectorChord-bm25's Block-WAND query path.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h).  Every function cites the reference
// file:line it follows (paths relative to /root/reference).  Nothing here is
// copied from the reference: the Rust code is generic over PostgreSQL pages and
// SIMD lanes; this file restates the *behaviour* over flat arrays.
//
// The storage layer (8 KiB pages, tapes, address trees: tuples.rs, tape.rs,
// address_*.rs) is not restated: bm25::search only uses it to fetch the values
// held here in flat arrays, in the same order.
//
// Build: see oracle/Makefile (g++ -O2, no -ffast-math: f64 results must be
// IEEE-exact, the same as rustc's).

#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// bm25.rs:15-283  fieldnorm <-> length
// The 256-entry table is Lucene's SmallFloat.byte4ToInt (24 "free" values, then
// a 4-bit-mantissa float): regenerated from that published rule instead of
// being copied; tests/test_oracle_pins.py checks it against the reference
// table when /root/reference is present.
// ---------------------------------------------------------------------------
// blocks the Block-WAND restatement decompressed on this thread (tools/wand_pruning.py: how much the reference's own
// traversal skips on a query shape)
static thread_local unsigned long long g_blocks_decoded = 0;

struct FieldnormTable {
    uint32_t v[256];
    FieldnormTable() {
        for (uint32_t i = 0; i < 256; ++i) {
            if (i < 24) {
                v[i] = i;
            } else {
                uint32_t j = i - 24, bits = j & 7;
                int shift = int(j >> 3) - 1;
                uint64_t dec = shift < 0 ? bits : (uint64_t(bits | 8) << shift);
                v[i] = uint32_t(24 + dec);
            }
        }
    }
};
const FieldnormTable g_fn;

inline uint32_t fieldnorm_to_length(uint8_t f) { return g_fn.v[f]; }  // bm25.rs:274-276

// bm25.rs:278-283: binary_search; Ok(i) -> i, Err(i) -> i - 1 (largest entry <= length).
inline uint8_t length_to_fieldnorm(uint32_t length) {
    const uint32_t *p = std::upper_bound(g_fn.v, g_fn.v + 256, length);
    return uint8_t((p - g_fn.v) - 1);
}

// bm25.rs:285-289
inline double idf(uint32_t n_docs, uint32_t df) {
    double n = double(n_docs), d = double(df);
    return std::log((n + 1.0) / (d + 0.5));
}

// bm25.rs:291-295
inline double tf_norm(uint8_t fieldnorm, uint32_t tf, double k1, double b, double avgdl) {
    double t = double(tf);
    double dl = double(fieldnorm_to_length(fieldnorm));
    return (t * (k1 + 1.0)) / (t + k1 * (1.0 - b + b * dl / avgdl));
}

// bm25.rs:297-332
struct Wand {
    double tf = 0.0;
    uint8_t fieldnorm = 255;
    uint32_t term_frequency = 0;
    void push(uint8_t fn, uint32_t t, double k1, double b, double avgdl) {
        double x = tf_norm(fn, t, k1, b, avgdl);
        if (tf < x) {
            tf = x;
            fieldnorm = fn;
            term_frequency = t;
        }
    }
    void extend(const Wand &o) {
        if (tf < o.tf) {
            tf = o.tf;
            fieldnorm = o.fieldnorm;
            term_frequency = o.term_frequency;
        }
    }
};

// bm25.rs:334-359.  The reference rebuilds the 256-entry s1 table per term;
// so does this (it is part of the per-query CPU cost being timed).
struct Cache {
    double s0;
    double s1[256];
    Cache(uint32_t n_docs, uint32_t df, double k1, double b, double avgdl) {
        s0 = idf(n_docs, df) * (k1 + 1.0);
        for (int f = 0; f < 256; ++f) {
            double dl = double(fieldnorm_to_length(uint8_t(f)));
            s1[f] = k1 * (1.0 - b + b * dl / avgdl);
        }
    }
    double evaluate(uint8_t fieldnorm, uint32_t tf) const {
        double t = double(tf);
        return (t * s0) / (t + s1[fieldnorm]);
    }
};

// ---------------------------------------------------------------------------
// crates/score/src/lib.rs:46-60
// ---------------------------------------------------------------------------
inline int64_t score_from_f64(double v) {
    int64_t bits;
    std::memcpy(&bits, &v, 8);
    uint64_t mask = (uint64_t)(bits >> 63) >> 1;
    return bits ^ (int64_t)mask;
}
inline double score_to_f64(int64_t s) {
    uint64_t mask = (uint64_t)(s >> 63) >> 1;
    int64_t bits = s ^ (int64_t)mask;
    double v;
    std::memcpy(&v, &bits, 8);
    return v;
}

// ---------------------------------------------------------------------------
// Codec.  crates/simd/src/bitpacking.rs:15-98 (the compress!/decompress! macros),
// bitpacking_u32_ordered.rs:15-34,82-125,190-237, bitpacking_u32_unordered.rs:15-32,
// bytepacking_u32_{ordered,unordered}.rs.
//
// Bit-packed layout (little endian): value i (0..127) belongs to lane l = i % 4
// at step t = i / 4.  Each lane is an LSB-first stream of 32 fields of `b`
// bits; 32-bit word w of lane l is stored at byte 16*w + 4*l.
// ---------------------------------------------------------------------------
inline uint8_t bits_needed(uint32_t reduce_or) {
    return reduce_or ? uint8_t(32 - __builtin_clz(reduce_or)) : 0;  // 1 + ilog2
}

inline void put_le32(uint8_t *p, uint32_t v) {
    p[0] = uint8_t(v);
    p[1] = uint8_t(v >> 8);
    p[2] = uint8_t(v >> 16);
    p[3] = uint8_t(v >> 24);
}
inline uint32_t get_le32(const uint8_t *p) {
    return uint32_t(p[0]) | uint32_t(p[1]) << 8 | uint32_t(p[2]) << 16 | uint32_t(p[3]) << 24;
}

// fields[128] (already deltas for the ordered flavour) -> 16*b bytes
void bitpack128(uint8_t b, const uint32_t *fields, uint8_t *out) {
    if (b == 0) return;  // bitpacking_u32_ordered.rs:117 "0 => ()"
    if (b == 32) {       // raw copy, handled by callers (absolute values)
        for (int i = 0; i < 128; ++i) put_le32(out + 4 * i, fields[i]);
        return;
    }
    uint32_t words[32][4];
    std::memset(words, 0, sizeof words);
    for (int i = 0; i < 128; ++i) {
        int l = i & 3, t = i >> 2;
        int bit = t * b, w = bit >> 5, sh = bit & 31;
        uint32_t v = fields[i];
        words[w][l] |= v << sh;
        if (sh + b > 32) words[w + 1][l] |= v >> (32 - sh);
    }
    for (int w = 0; w < b; ++w)
        for (int l = 0; l < 4; ++l) put_le32(out + 16 * w + 4 * l, words[w][l]);
}

void bitunpack128(uint8_t b, const uint8_t *in, uint32_t *fields) {
    uint32_t mask = (b >= 32) ? 0xffffffffu : ((1u << b) - 1);
    for (int i = 0; i < 128; ++i) {
        int l = i & 3, t = i >> 2;
        int bit = t * b, w = bit >> 5, sh = bit & 31;
        uint32_t v = (get_le32(in + 16 * w + 4 * l) >> sh) & mask;
        if (sh + b > 32) v |= (get_le32(in + 16 * (w + 1) + 4 * l) << (32 - sh)) & mask;
        fields[i] = v;
    }
}

// compression.rs:36-63
uint8_t compress_document_ids(uint32_t min_doc, const uint32_t *ids, uint32_t n, uint8_t *out,
                              uint32_t *out_len) {
    uint32_t last = min_doc, reduce_or = 0;
    for (uint32_t i = 0; i < n; ++i) {
        reduce_or |= ids[i] - last;
        last = ids[i];
    }
    uint8_t bits = bits_needed(reduce_or);
    if (n == 128) {
        *out_len = 16u * bits;
        if (bits == 32) {  // bitpacking_u32_ordered.rs:119-121: raw absolute ids
            bitpack128(32, ids, out);
        } else {
            uint32_t d[128];
            last = min_doc;
            for (int i = 0; i < 128; ++i) {
                d[i] = ids[i] - last;
                last = ids[i];
            }
            bitpack128(bits, d, out);
        }
        return bits;  // flag bit 7 = 0
    }
    uint8_t w = std::max<uint8_t>(1, uint8_t((bits + 7) / 8));  // bytepacking_u32_ordered.rs:17-30
    *out_len = w * n;
    last = min_doc;
    for (uint32_t i = 0; i < n; ++i) {
        // w == 4: raw absolute (bytepacking_u32_ordered.rs:195); else delta bytes
        uint32_t v = (w == 4) ? ids[i] : ids[i] - last;
        last = ids[i];
        for (uint8_t j = 0; j < w; ++j) out[i * w + j] = uint8_t(v >> (8 * j));
    }
    return uint8_t(0x80 | w);
}

// compression.rs:94-110
uint8_t compress_term_frequencies(const uint32_t *tfs, uint32_t n, uint8_t *out,
                                  uint32_t *out_len) {
    uint32_t reduce_or = 0;
    for (uint32_t i = 0; i < n; ++i) reduce_or |= tfs[i];
    uint8_t bits = bits_needed(reduce_or);
    if (n == 128) {
        *out_len = 16u * bits;
        bitpack128(bits, tfs, out);
        return bits;
    }
    uint8_t w = std::max<uint8_t>(1, uint8_t((bits + 7) / 8));
    *out_len = w * n;
    for (uint32_t i = 0; i < n; ++i)
        for (uint8_t j = 0; j < w; ++j) out[i * w + j] = uint8_t(tfs[i] >> (8 * j));
    return uint8_t(0x80 | w);
}

// compression.rs:65-92.  `out` keeps stale contents when bitwidth == 0, like the
// reference's Decompressed buffer (bitpacking_u32_ordered.rs:228-229).
uint32_t decompress_document_ids(uint32_t min_doc, uint8_t meta, const uint8_t *in,
                                 uint32_t in_len, uint32_t *out) {
    if ((meta >> 7) == 0) {
        uint8_t b = meta & 127;
        if (b == 0) return 128;
        if (b == 32) {
            for (int i = 0; i < 128; ++i) out[i] = get_le32(in + 4 * i);
            return 128;
        }
        bitunpack128(b, in, out);
        uint32_t state = min_doc;  // bitpacking_u32_ordered.rs:191-218: running sum in index order
        for (int i = 0; i < 128; ++i) {
            state += out[i];
            out[i] = state;
        }
        return 128;
    }
    uint8_t w = meta & 127;
    uint32_t n = in_len / w;
    uint32_t state = min_doc;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = 0;
        for (uint8_t j = 0; j < w; ++j) v |= uint32_t(in[i * w + j]) << (8 * j);
        if (w == 4) {
            out[i] = v;  // bytepacking_u32_ordered.rs:211 raw copy
        } else {
            state += v;
            out[i] = state;
        }
    }
    return n;
}

// compression.rs:112-136
uint32_t decompress_term_frequencies(uint8_t meta, const uint8_t *in, uint32_t in_len,
                                     uint32_t *out) {
    if ((meta >> 7) == 0) {
        uint8_t b = meta & 127;
        if (b == 0) return 128;
        if (b == 32) {
            for (int i = 0; i < 128; ++i) out[i] = get_le32(in + 4 * i);
            return 128;
        }
        bitunpack128(b, in, out);
        return 128;
    }
    uint8_t w = meta & 127;
    uint32_t n = in_len / w;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t v = 0;
        for (uint8_t j = 0; j < w; ++j) v |= uint32_t(in[i * w + j]) << (8 * j);
        out[i] = v;
    }
    return n;
}

// ---------------------------------------------------------------------------
// Rust std::collections::BinaryHeap, restated from the documented std
// algorithm (NOT in the reference tree; "assumed").
// Max-heap on `Cmp`; Cmp::le(a,b) is Rust's `a <= b`.
// ---------------------------------------------------------------------------
template <class T, class Cmp>
struct RustHeap {
    std::vector<T> data;

    static bool le(const T &a, const T &b) { return Cmp::le(a, b); }
    static bool ge(const T &a, const T &b) { return Cmp::le(b, a); }
    static bool lt(const T &a, const T &b) { return !Cmp::le(b, a); }

    size_t len() const { return data.size(); }
    bool empty() const { return data.empty(); }
    T *peek() { return data.empty() ? nullptr : &data[0]; }

    void sift_up(size_t start, size_t pos) {
        T elem = std::move(data[pos]);
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (le(elem, data[parent])) break;
            data[pos] = std::move(data[parent]);
            pos = parent;
        }
        data[pos] = std::move(elem);
    }
    void sift_down_range(size_t pos, size_t end) {
        T elem = std::move(data[pos]);
        size_t child = 2 * pos + 1;
        size_t lim = end >= 2 ? end - 2 : 0;  // end.saturating_sub(2)
        while (child <= lim && end >= 2) {
            if (le(data[child], data[child + 1])) child += 1;
            if (ge(elem, data[child])) {
                data[pos] = std::move(elem);
                return;
            }
            data[pos] = std::move(data[child]);
            pos = child;
            child = 2 * pos + 1;
        }
        if (end >= 1 && child == end - 1 && lt(elem, data[child])) {
            data[pos] = std::move(data[child]);
            pos = child;
        }
        data[pos] = std::move(elem);
    }
    void sift_down_to_bottom(size_t pos) {
        size_t end = data.size(), start = pos;
        T elem = std::move(data[pos]);
        size_t child = 2 * pos + 1;
        size_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim && end >= 2) {
            if (le(data[child], data[child + 1])) child += 1;
            data[pos] = std::move(data[child]);
            pos = child;
            child = 2 * pos + 1;
        }
        if (end >= 1 && child == end - 1) {
            data[pos] = std::move(data[child]);
            pos = child;
        }
        data[pos] = std::move(elem);
        sift_up(start, pos);
    }
    void push(T x) {
        size_t old = data.size();
        data.push_back(std::move(x));
        sift_up(0, old);
    }
    bool pop(T &out) {
        if (data.empty()) return false;
        T item = std::move(data.back());
        data.pop_back();
        if (!data.empty()) {
            std::swap(item, data[0]);
            sift_down_to_bottom(0);
        }
        out = std::move(item);
        return true;
    }
    void rebuild() {  // BinaryHeap::from(vec)
        size_t n = data.size() / 2;
        while (n > 0) {
            n -= 1;
            sift_down_range(n, data.size());
        }
    }
    std::vector<T> into_sorted_vec() {
        size_t end = data.size();
        while (end > 1) {
            end -= 1;
            std::swap(data[0], data[end]);
            sift_down_range(0, end);
        }
        return std::move(data);
    }
};

// ---------------------------------------------------------------------------
// Flattened index (DESIGN.md "Index layout"): what flush.rs writes to
// Token/Summary/Block/Document tapes, kept as arrays.
// ---------------------------------------------------------------------------
}  // namespace

struct orc_index {
    uint32_t n_docs = 0, n_terms = 0, n_blocks = 0;
    uint64_t sum_len = 0;
    double k1 = 1.2, b = 0.75;
    std::vector<uint8_t> term_key;
    std::vector<uint32_t> term_df;
    std::vector<uint8_t> term_wand_fn;
    std::vector<uint32_t> term_wand_tf;
    std::vector<uint32_t> term_first_block;
    std::vector<uint32_t> blk_min_doc, blk_max_doc;
    std::vector<uint8_t> blk_n, blk_wand_fn;
    std::vector<uint32_t> blk_wand_tf;
    std::vector<uint8_t> blk_meta_doc, blk_meta_tf;
    std::vector<uint32_t> blk_off8;
    std::vector<uint8_t> blob;
    std::vector<uint8_t> doc_fieldnorm;
    std::vector<uint16_t> doc_payload;

    double avgdl() const { return double(sum_len) / double(n_docs); }  // search.rs:49-51
    uint32_t doc_bytes(uint32_t blk) const {
        uint8_t m = blk_meta_doc[blk];
        return (m >> 7) ? uint32_t(m & 127) * blk_n[blk] : 16u * (m & 127);
    }
    uint32_t tf_bytes(uint32_t blk) const {
        uint8_t m = blk_meta_tf[blk];
        return (m >> 7) ? uint32_t(m & 127) * blk_n[blk] : 16u * (m & 127);
    }
    const uint8_t *doc_ptr(uint32_t blk) const { return blob.data() + 8ull * blk_off8[blk]; }
    const uint8_t *tf_ptr(uint32_t blk) const {
        return doc_ptr(blk) + ((doc_bytes(blk) + 7u) & ~7u);  // tuples.rs:1003-1005 pad to 8
    }
};

namespace {

// flush.rs:40-158
orc_index *build_index(double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                       const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                       const uint64_t *term_start, const uint32_t *post_doc,
                       const uint32_t *post_tf) {
    auto ix = std::make_unique<orc_index>();
    ix->k1 = k1;
    ix->b = b;
    ix->n_docs = n_docs;
    ix->doc_fieldnorm.resize(n_docs);
    ix->doc_payload.assign(doc_payload, doc_payload + 3ull * n_docs);
    for (uint32_t d = 0; d < n_docs; ++d) {  // flush.rs:54-65
        ix->sum_len += doc_len[d];
        ix->doc_fieldnorm[d] = length_to_fieldnorm(doc_len[d]);
    }
    double avgdl = double(ix->sum_len) / double(n_docs);  // flush.rs:67

    ix->term_first_block.push_back(0);
    ix->blk_off8.push_back(0);
    for (uint32_t t = 0; t < n_terms; ++t) {
        uint64_t s = term_start[t], e = term_start[t + 1];
        if (s == e) continue;  // a token only exists if it has mappings (flush.rs:74)
        Wand token_wand;
        for (uint64_t p = s; p < e; p += 128) {  // flush.rs:79-90 block cut
            uint32_t n = uint32_t(std::min<uint64_t>(128, e - p));
            const uint32_t *ids = post_doc + p, *tfs = post_tf + p;
            uint8_t buf[512];
            uint32_t len = 0;
            uint8_t meta_doc = compress_document_ids(ids[0], ids, n, buf, &len);
            size_t at = ix->blob.size();
            ix->blob.insert(ix->blob.end(), buf, buf + len);
            while (ix->blob.size() % 8) ix->blob.push_back(0);
            uint8_t meta_tf = compress_term_frequencies(tfs, n, buf, &len);
            ix->blob.insert(ix->blob.end(), buf, buf + len);
            while (ix->blob.size() % 8) ix->blob.push_back(0);
            (void)at;
            Wand block_wand;  // flush.rs:101-110
            for (uint32_t i = 0; i < n; ++i)
                block_wand.push(ix->doc_fieldnorm[ids[i]], tfs[i], k1, b, avgdl);
            token_wand.extend(block_wand);  // flush.rs:112
            ix->blk_min_doc.push_back(ids[0]);
            ix->blk_max_doc.push_back(ids[n - 1]);
            ix->blk_n.push_back(uint8_t(n));
            ix->blk_wand_fn.push_back(block_wand.fieldnorm);
            ix->blk_wand_tf.push_back(block_wand.term_frequency);
            ix->blk_meta_doc.push_back(meta_doc);
            ix->blk_meta_tf.push_back(meta_tf);
            ix->blk_off8.push_back(uint32_t(ix->blob.size() / 8));
        }
        ix->term_key.insert(ix->term_key.end(), term_key + 16ull * t, term_key + 16ull * t + 16);
        ix->term_df.push_back(uint32_t(e - s));
        ix->term_wand_fn.push_back(token_wand.fieldnorm);
        ix->term_wand_tf.push_back(token_wand.term_frequency);
        ix->term_first_block.push_back(uint32_t(ix->blk_n.size()));
    }
    ix->n_terms = uint32_t(ix->term_df.size());
    ix->n_blocks = uint32_t(ix->blk_n.size());
    return ix.release();
}

// ---------------------------------------------------------------------------
// search.rs:284-314  Results
// ---------------------------------------------------------------------------
struct ResItem {
    int64_t score;  // Score key
    uint32_t doc;   // AlwaysEqual payload: never takes part in comparisons
    uint16_t payload[3];
};
struct ResCmp {
    // element = (Reverse<Score>, AlwaysEqual): a <= b  <=>  a.score >= b.score
    static bool le(const ResItem &a, const ResItem &b) { return a.score >= b.score; }
};
struct Results {
    size_t limit;
    int64_t threshold;
    RustHeap<ResItem, ResCmp> internal;
    Results(size_t k, double thr) : limit(k), threshold(score_from_f64(thr)) {}
    double thr() const { return score_to_f64(threshold); }
    void push(double key, uint32_t doc, const uint16_t *payload) {
        ResItem it{score_from_f64(key), doc, {payload[0], payload[1], payload[2]}};
        internal.push(it);
        if (internal.len() > limit) {
            ResItem tmp;
            internal.pop(tmp);
        }
        if (internal.len() == limit)
            threshold = std::max(threshold, internal.peek()->score);
    }
};

// ---------------------------------------------------------------------------
// search.rs:316-496  Cursor over one token's summaries / blocks
// ---------------------------------------------------------------------------
struct Summary {
    uint32_t min_doc, max_doc;
    uint8_t n, wand_fn;
    uint32_t wand_tf;
    uint32_t blk;  // stands for wptr_block
};

struct Cursor {
    const orc_index *ix;
    Cache bm25;
    double token_upper_bound;
    uint32_t document_id;
    uint8_t position_in_block;
    uint32_t next_blk, end_blk;  // TruncatedTapeReader: exactly ceil(df/128) summaries
    Summary summary;
    double block_upper_bound;
    bool filled;
    uint32_t docs[128], tfs[128];
    uint32_t n_docs_dec = 0, n_tfs_dec = 0;

    Summary next_summary() {  // search.rs:484-496
        if (next_blk < end_blk) {
            uint32_t j = next_blk++;
            return Summary{ix->blk_min_doc[j], ix->blk_max_doc[j], ix->blk_n[j],
                           ix->blk_wand_fn[j], ix->blk_wand_tf[j], j};
        }
        return Summary{UINT32_MAX, UINT32_MAX, 1, 255, 0, UINT32_MAX};
    }
    Cursor(const orc_index *ix_, uint32_t term, const Cache &c) : ix(ix_), bm25(c) {  // 352-396
        std::memset(docs, 0, sizeof docs);
        std::memset(tfs, 0, sizeof tfs);
        token_upper_bound = bm25.evaluate(ix->term_wand_fn[term], ix->term_wand_tf[term]);
        next_blk = ix->term_first_block[term];
        end_blk = next_blk + (ix->term_df[term] + 127) / 128;
        summary = next_summary();
        block_upper_bound = bm25.evaluate(summary.wand_fn, summary.wand_tf);
        document_id = summary.min_doc;
        position_in_block = 0;
        filled = false;
    }
    void fill_block() {  // search.rs:498-518
        ++g_blocks_decoded;
        uint32_t j = summary.blk;
        n_docs_dec = decompress_document_ids(summary.min_doc, ix->blk_meta_doc[j], ix->doc_ptr(j),
                                             ix->doc_bytes(j), docs);
        n_tfs_dec = decompress_term_frequencies(ix->blk_meta_tf[j], ix->tf_ptr(j),
                                                ix->tf_bytes(j), tfs);
        filled = true;
    }
    void seek_block(uint32_t target) {  // search.rs:412-431
        if (target <= summary.max_doc) return;
        while (summary.max_doc < target) summary = next_summary();
        document_id = summary.min_doc;
        position_in_block = 0;
        block_upper_bound = bm25.evaluate(summary.wand_fn, summary.wand_tf);
        filled = false;
    }
    void seek(uint32_t target) {  // search.rs:432-466
        seek_block(target);
        if (target <= document_id) return;
        if (target == summary.max_doc) {
            document_id = summary.max_doc;
            position_in_block = uint8_t(summary.n - 1);
            return;
        }
        if (!filled) fill_block();
        uint32_t i;
        if (target == document_id + 1) {
            i = uint32_t(position_in_block) + 1;
        } else {
            uint32_t start = uint32_t(position_in_block) + 1;
            // slice::binary_search: Ok(pos) or Err(insertion point) = first >= target
            const uint32_t *lo = std::lower_bound(docs + start, docs + n_docs_dec, target);
            i = uint32_t(lo - docs);
        }
        document_id = docs[i];
        position_in_block = uint8_t(i);
    }
    uint32_t get() {  // search.rs:467-481
        if (!filled) fill_block();
        return tfs[position_in_block];
    }
};

using CursorBox = std::unique_ptr<Cursor>;
struct CursorCmp {
    // Ord for Cursor is reversed on document_id (search.rs:331-349):
    // a <= b  <=>  cmp(b.doc, a.doc) != Greater  <=>  b.doc <= a.doc
    static bool le(const CursorBox &a, const CursorBox &b) {
        return b->document_id <= a->document_id;
    }
};

struct GrowingDocs {
    uint32_t n = 0;
    const uint64_t *start = nullptr;
    const uint32_t *term = nullptr, *tf = nullptr;
    const uint8_t *fieldnorm = nullptr;
    const uint16_t *payload = nullptr;
    const uint8_t *deleted = nullptr;
};

// search.rs:28-282 with filter == |_| true
uint32_t search_wand(const orc_index *ix, const uint32_t *terms, uint32_t n_terms, uint32_t k,
                     const GrowingDocs *grow, orc_hit *out) {
    if (k == 0) return 0;
    const double avgdl = ix->avgdl();

    // search.rs:53-79: tokens found in the index, in query (ascending key) order
    std::vector<uint32_t> tok_term;
    std::vector<Cache> tok_cache;
    for (uint32_t i = 0; i < n_terms; ++i) {
        if (terms[i] >= ix->n_terms) continue;  // address_tokens::read -> None
        tok_term.push_back(terms[i]);
        tok_cache.emplace_back(ix->n_docs, ix->term_df[terms[i]], ix->k1, ix->b, avgdl);
    }

    Results results(k, 0.0);

    // search.rs:83-135 growing segment: scored first, seeds the threshold
    if (grow) {
        for (uint32_t g = 0; g < grow->n; ++g) {
            if (grow->deleted && grow->deleted[g]) continue;
            double result = 0.0;
            for (uint64_t p = grow->start[g]; p < grow->start[g + 1]; ++p) {
                auto it = std::lower_bound(tok_term.begin(), tok_term.end(), grow->term[p]);
                if (it != tok_term.end() && *it == grow->term[p])
                    result += tok_cache[it - tok_term.begin()].evaluate(grow->fieldnorm[g],
                                                                        grow->tf[p]);
            }
            if (results.thr() < result) results.push(result, UINT32_MAX - g, grow->payload + 3ull * g);
        }
    }

    RustHeap<CursorBox, CursorCmp> head;
    for (size_t i = 0; i < tok_term.size(); ++i)  // search.rs:137-147
        head.data.push_back(std::make_unique<Cursor>(ix, tok_term[i], tok_cache[i]));
    head.rebuild();  // BinaryHeap::from(cursors)
    std::vector<CursorBox> tail;

    for (;;) {  // 'main
        CursorBox lead0;
        {   // search.rs:152-169 pivot selection
            double sum = 0.0;
            for (auto &c : tail) sum += c->token_upper_bound;
            bool done = true;
            CursorBox cur;
            while (head.pop(cur)) {
                if (cur->document_id == UINT32_MAX) {
                    done = true;
                    break;
                }
                if (results.thr() < sum + cur->token_upper_bound) {
                    lead0 = std::move(cur);
                    done = false;
                    break;
                }
                sum += cur->token_upper_bound;
                tail.push_back(std::move(cur));
            }
            if (done) break;
        }
        const uint32_t document_id = lead0->document_id;
        std::vector<CursorBox> lead;
        lead.push_back(std::move(lead0));
        while (head.peek() && (*head.peek())->document_id == document_id) {  // 171-176
            CursorBox c;
            head.pop(c);
            lead.push_back(std::move(c));
        }
        {   // search.rs:177-192: extract_if, ALL overshooting tail cursors drained
            bool failed = false;
            std::vector<CursorBox> keep;
            std::vector<CursorBox> failures;
            for (auto &c : tail) {
                c->seek_block(document_id);
                if (document_id < c->document_id) {
                    failed = true;
                    failures.push_back(std::move(c));
                } else {
                    keep.push_back(std::move(c));
                }
            }
            if (failed) {
                tail = std::move(keep);
                for (auto &c : lead) head.push(std::move(c));
                for (auto &c : failures) head.push(std::move(c));
                continue;
            }
            tail = std::move(keep);
        }
        double sum_block_ub = 0.0;  // search.rs:193-202
        for (auto &c : tail) sum_block_ub += c->block_upper_bound;
        for (auto &c : lead) sum_block_ub += c->block_upper_bound;
        if (results.thr() < sum_block_ub) {
            {   // search.rs:204-216: lazy extract_if, only the FIRST failure is
                // taken; cursors after it are not seeked in this round
                size_t fail_at = tail.size();
                for (size_t i = 0; i < tail.size(); ++i) {
                    tail[i]->seek(document_id);
                    if (document_id < tail[i]->document_id) {
                        fail_at = i;
                        break;
                    }
                }
                if (fail_at != tail.size()) {
                    CursorBox failure = std::move(tail[fail_at]);
                    tail.erase(tail.begin() + fail_at);
                    for (auto &c : lead) head.push(std::move(c));
                    head.push(std::move(failure));
                    continue;
                }
            }
            // search.rs:217-229: DocumentTuple lookup (fieldnorm, payload)
            uint8_t fieldnorm = ix->doc_fieldnorm[document_id];
            const uint16_t *payload = ix->doc_payload.data() + 3ull * document_id;
            {   // filter(payload) == true; sum over tail then lead (search.rs:231-236)
                double result = 0.0;
                for (auto &c : tail) result += c->bm25.evaluate(fieldnorm, c->get());
                for (auto &c : lead) result += c->bm25.evaluate(fieldnorm, c->get());
                results.push(result, document_id, payload);
            }
            for (auto &c : tail) {  // search.rs:238-241
                c->seek(1 + document_id);
                head.push(std::move(c));
            }
            for (auto &c : lead) {
                c->seek(1 + document_id);
                head.push(std::move(c));
            }
            tail.clear();
        } else {  // search.rs:243-279
            uint32_t min_bm = UINT32_MAX;
            for (auto &c : lead) min_bm = std::min(min_bm, c->summary.max_doc);
            for (auto &c : tail) min_bm = std::min(min_bm, c->summary.max_doc);
            uint32_t head_doc = head.peek() ? (*head.peek())->document_id : UINT32_MAX;
            uint32_t seek_doc = std::min(1 + min_bm, head_doc);
            double mx = -INFINITY;
            int which = 0;
            size_t at = 0;
            for (size_t j = 0; j < lead.size(); ++j)
                if (lead[j]->token_upper_bound > mx) {
                    mx = lead[j]->token_upper_bound;
                    which = 0;
                    at = j;
                }
            for (size_t j = 0; j < tail.size(); ++j)
                if (tail[j]->token_upper_bound > mx) {
                    mx = tail[j]->token_upper_bound;
                    which = 1;
                    at = j;
                }
            std::vector<CursorBox> &src = which == 0 ? lead : tail;
            CursorBox c = std::move(src[at]);
            src.erase(src.begin() + at);
            c->seek(seek_doc);
            head.push(std::move(c));
            for (auto &l : lead) head.push(std::move(l));
        }
    }
    std::vector<ResItem> sorted = results.internal.into_sorted_vec();  // descending score
    for (size_t i = 0; i < sorted.size(); ++i) {
        out[i].score = score_to_f64(sorted[i].score);
        out[i].doc_id = sorted[i].doc;
        std::memcpy(out[i].payload, sorted[i].payload, 6);
        out[i]._pad = 0;
    }
    return uint32_t(sorted.size());
}

// Canonical brute force: what `evaluate` over every document would give if it
// used the index path's Cache arithmetic, summed in ascending key order.
uint32_t search_brute(const orc_index *ix, const uint32_t *terms, uint32_t n_terms, uint32_t k,
                      orc_hit *out) {
    if (k == 0) return 0;
    const double avgdl = ix->avgdl();
    // one accumulator per thread, kept across calls and wiped through the touched list: allocating and
    // zeroing n_docs doubles per query made a full-batch check on 10 M documents take minutes
    static thread_local std::vector<double> acc;
    if (acc.size() < ix->n_docs) acc.assign(ix->n_docs, 0.0);
    std::vector<uint32_t> touched;
    uint32_t docs[128], tfs[128];
    for (uint32_t i = 0; i < n_terms; ++i) {
        uint32_t t = terms[i];
        if (t >= ix->n_terms) continue;
        Cache c(ix->n_docs, ix->term_df[t], ix->k1, ix->b, avgdl);
        for (uint32_t j = ix->term_first_block[t]; j < ix->term_first_block[t + 1]; ++j) {
            uint32_t n = decompress_document_ids(ix->blk_min_doc[j], ix->blk_meta_doc[j],
                                                 ix->doc_ptr(j), ix->doc_bytes(j), docs);
            decompress_term_frequencies(ix->blk_meta_tf[j], ix->tf_ptr(j), ix->tf_bytes(j), tfs);
            for (uint32_t p = 0; p < n; ++p) {
                uint32_t d = docs[p];
                if (acc[d] == 0.0) touched.push_back(d);
                acc[d] += c.evaluate(ix->doc_fieldnorm[d], tfs[p]);
            }
        }
    }
    auto better = [&](uint32_t a, uint32_t b) {
        return acc[a] > acc[b] || (acc[a] == acc[b] && a < b);
    };
    size_t kk = std::min<size_t>(k, touched.size());
    std::partial_sort(touched.begin(), touched.begin() + kk, touched.end(), better);
    for (size_t i = 0; i < kk; ++i) {
        uint32_t d = touched[i];
        out[i].score = acc[d];
        out[i].doc_id = d;
        std::memcpy(out[i].payload, ix->doc_payload.data() + 3ull * d, 6);
        out[i]._pad = 0;
    }
    for (uint32_t d : touched) acc[d] = 0.0;
    return uint32_t(kk);
}

#include "dense_model.inc"

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

uint32_t orc_fieldnorm_to_length(uint8_t f) { return fieldnorm_to_length(f); }
uint8_t orc_length_to_fieldnorm(uint32_t l) { return length_to_fieldnorm(l); }
double orc_idf(uint32_t n, uint32_t df) { return idf(n, df); }
double orc_tf(uint8_t f, uint32_t tf, double k1, double b, double avgdl) {
    return tf_norm(f, tf, k1, b, avgdl);
}
double orc_cache_evaluate(uint32_t n, uint32_t df, double k1, double b, double avgdl, uint8_t f,
                          uint32_t tf) {
    return Cache(n, df, k1, b, avgdl).evaluate(f, tf);
}
int64_t orc_score_from_f64(double v) { return score_from_f64(v); }
double orc_score_to_f64(int64_t s) { return score_to_f64(s); }

uint8_t orc_compress_document_ids(uint32_t min_doc, const uint32_t *ids, uint32_t n, uint8_t *out,
                                  uint32_t *out_len) {
    return compress_document_ids(min_doc, ids, n, out, out_len);
}
uint8_t orc_compress_term_frequencies(const uint32_t *tfs, uint32_t n, uint8_t *out,
                                      uint32_t *out_len) {
    return compress_term_frequencies(tfs, n, out, out_len);
}
uint32_t orc_decompress_document_ids(uint32_t min_doc, uint8_t meta, const uint8_t *in,
                                     uint32_t in_len, uint32_t *out) {
    return decompress_document_ids(min_doc, meta, in, in_len, out);
}
uint32_t orc_decompress_term_frequencies(uint8_t meta, const uint8_t *in, uint32_t in_len,
                                         uint32_t *out) {
    return decompress_term_frequencies(meta, in, in_len, out);
}

uint32_t orc_heap_script(const int64_t *keys, const int32_t *ops, uint32_t n_ops,
                         int32_t *popped_tags, int32_t *sorted_tags, uint32_t *n_sorted) {
    struct It {
        int64_t key;
        int32_t tag;
    };
    struct C {
        static bool le(const It &a, const It &b) { return a.key <= b.key; }
    };
    RustHeap<It, C> h;
    uint32_t np = 0;
    for (uint32_t i = 0; i < n_ops; ++i) {
        if (ops[i] >= 0) {
            h.push(It{keys[i], int32_t(i)});
        } else {
            It it;
            if (h.pop(it)) popped_tags[np++] = it.tag;
        }
    }
    auto v = h.into_sorted_vec();
    for (size_t i = 0; i < v.size(); ++i) sorted_tags[i] = v[i].tag;
    *n_sorted = uint32_t(v.size());
    return np;
}

orc_index *orc_index_build(double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                           const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                           const uint64_t *term_start, const uint32_t *post_doc,
                           const uint32_t *post_tf) {
    return build_index(k1, b, n_docs, doc_len, doc_payload, n_terms, term_key, term_start,
                       post_doc, post_tf);
}

orc_index *orc_index_from_view(const orc_index_view *v) {
    auto ix = std::make_unique<orc_index>();
    ix->n_docs = v->n_docs;
    ix->n_terms = v->n_terms;
    ix->n_blocks = v->n_blocks;
    ix->sum_len = v->sum_len;
    ix->k1 = v->k1;
    ix->b = v->b;
    ix->term_key.assign(v->term_key, v->term_key + 16ull * v->n_terms);
    ix->term_df.assign(v->term_df, v->term_df + v->n_terms);
    ix->term_wand_fn.assign(v->term_wand_fn, v->term_wand_fn + v->n_terms);
    ix->term_wand_tf.assign(v->term_wand_tf, v->term_wand_tf + v->n_terms);
    ix->term_first_block.assign(v->term_first_block, v->term_first_block + v->n_terms + 1);
    ix->blk_min_doc.assign(v->blk_min_doc, v->blk_min_doc + v->n_blocks);
    ix->blk_max_doc.assign(v->blk_max_doc, v->blk_max_doc + v->n_blocks);
    ix->blk_n.assign(v->blk_n, v->blk_n + v->n_blocks);
    ix->blk_wand_fn.assign(v->blk_wand_fn, v->blk_wand_fn + v->n_blocks);
    ix->blk_wand_tf.assign(v->blk_wand_tf, v->blk_wand_tf + v->n_blocks);
    ix->blk_meta_doc.assign(v->blk_meta_doc, v->blk_meta_doc + v->n_blocks);
    ix->blk_meta_tf.assign(v->blk_meta_tf, v->blk_meta_tf + v->n_blocks);
    ix->blk_off8.assign(v->blk_off8, v->blk_off8 + v->n_blocks + 1);
    ix->blob.assign(v->blob, v->blob + v->blob_bytes);
    ix->doc_fieldnorm.assign(v->doc_fieldnorm, v->doc_fieldnorm + v->n_docs);
    ix->doc_payload.assign(v->doc_payload, v->doc_payload + 3ull * v->n_docs);
    return ix.release();
}

void orc_index_free(orc_index *ix) { delete ix; }

void orc_index_get_view(const orc_index *ix, orc_index_view *v) {
    std::memset(v, 0, sizeof *v);
    v->n_docs = ix->n_docs;
    v->n_terms = ix->n_terms;
    v->n_blocks = ix->n_blocks;
    v->sum_len = ix->sum_len;
    v->blob_bytes = ix->blob.size();
    v->k1 = ix->k1;
    v->b = ix->b;
    v->term_key = ix->term_key.data();
    v->term_df = ix->term_df.data();
    v->term_wand_fn = ix->term_wand_fn.data();
    v->term_wand_tf = ix->term_wand_tf.data();
    v->term_first_block = ix->term_first_block.data();
    v->blk_min_doc = ix->blk_min_doc.data();
    v->blk_max_doc = ix->blk_max_doc.data();
    v->blk_n = ix->blk_n.data();
    v->blk_wand_fn = ix->blk_wand_fn.data();
    v->blk_wand_tf = ix->blk_wand_tf.data();
    v->blk_meta_doc = ix->blk_meta_doc.data();
    v->blk_meta_tf = ix->blk_meta_tf.data();
    v->blk_off8 = ix->blk_off8.data();
    v->blob = ix->blob.data();
    v->doc_fieldnorm = ix->doc_fieldnorm.data();
    v->doc_payload = ix->doc_payload.data();
}

uint32_t orc_search_wand(const orc_index *ix, const uint32_t *terms, uint32_t n_terms, uint32_t k,
                         orc_hit *out) {
    return search_wand(ix, terms, n_terms, k, nullptr, out);
}

unsigned long long orc_wand_blocks_decoded(int reset) {
    const unsigned long long v = g_blocks_decoded;
    if (reset) g_blocks_decoded = 0;
    return v;
}

uint32_t orc_search_brute(const orc_index *ix, const uint32_t *terms, uint32_t n_terms,
                          uint32_t k, orc_hit *out) {
    return search_brute(ix, terms, n_terms, k, out);
}

uint32_t orc_search_wand_growing(const orc_index *ix, const uint32_t *terms, uint32_t n_terms,
                                 uint32_t k, uint32_t n_grow, const uint64_t *g_start,
                                 const uint32_t *g_term, const uint32_t *g_tf,
                                 const uint8_t *g_fieldnorm, const uint16_t *g_payload,
                                 const uint8_t *g_deleted, orc_hit *out) {
    GrowingDocs g{n_grow, g_start, g_term, g_tf, g_fieldnorm, g_payload, g_deleted};
    return search_wand(ix, terms, n_terms, k, &g, out);
}

double orc_search_batch(const orc_index *ix, const uint32_t *terms, const uint32_t *q_off,
                        uint32_t nq, uint32_t k, int mode, int threads, orc_hit *out,
                        uint32_t *n_hits) {
    if (threads < 1) threads = 1;
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
        for (;;) {
            uint32_t q = next.fetch_add(1);
            if (q >= nq) break;
            const uint32_t *t = terms + q_off[q];
            uint32_t nt = q_off[q + 1] - q_off[q];
            orc_hit *o = out + size_t(q) * k;
            n_hits[q] = mode == 0 ? search_wand(ix, t, nt, k, nullptr, o)
                                  : search_brute(ix, t, nt, k, o);
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; ++i) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// evaluate.rs:22-74 (uses idf * tf, not Cache)
int64_t orc_evaluate(const orc_index *ix, const uint32_t *doc_terms, const uint32_t *doc_tfs,
                     uint32_t n_doc_terms, const uint32_t *terms, uint32_t n_terms) {
    uint64_t len = 0;  // vector.rs:77-83 saturating sum
    for (uint32_t i = 0; i < n_doc_terms; ++i) len = std::min<uint64_t>(len + doc_tfs[i], UINT32_MAX);
    uint8_t fieldnorm = length_to_fieldnorm(uint32_t(len));
    double avgdl = ix->avgdl();
    size_t cursor = 0;
    double result = 0.0;
    for (uint32_t i = 0; i < n_terms; ++i) {
        uint32_t key = terms[i];
        while (cursor < n_doc_terms && doc_terms[cursor] < key) cursor++;
        if (!(cursor < n_doc_terms && doc_terms[cursor] == key)) continue;
        if (key >= ix->n_terms) continue;
        uint32_t tfv = doc_tfs[cursor];
        result += idf(ix->n_docs, ix->term_df[key]) * tf_norm(fieldnorm, tfv, ix->k1, ix->b, avgdl);
    }
    return score_from_f64(result);
}

// exhaustive-evaluation byte count of one query
uint64_t orc_query_bytes(const orc_index *ix, const uint32_t *terms, uint32_t n_terms,
                         uint32_t k) {
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n_terms; ++i) {
        uint32_t t = terms[i];
        if (t >= ix->n_terms) continue;
        uint32_t b0 = ix->term_first_block[t], b1 = ix->term_first_block[t + 1];
        bytes += 8ull * (ix->blk_off8[b1] - ix->blk_off8[b0]);  // payloads incl. pad8
        bytes += (16ull + 24ull) * (b1 - b0);  // BlockTuple header + SummaryTuple
        bytes += ix->term_df[t];               // one fieldnorm byte per posting
    }
    return bytes + 14ull * k;
}

/* model of the device's dense-window kernel (dense_model.inc): a test aid, not an oracle */
uint32_t orc_dense_model(const orc_index *ix, const uint32_t *terms, uint32_t n_terms, uint32_t k, uint32_t wmax,
                         uint32_t w0, uint32_t lo, uint32_t hi, int phases, orc_hit *out, uint64_t *stats8) {
    DenseModelStats st;
    const uint32_t n = dense_model(ix, terms, n_terms, k, wmax, w0, lo, hi, phases, out, &st);
    std::memcpy(stats8, &st, sizeof st);
    return n;
}

}  // extern "C"
