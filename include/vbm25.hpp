// vbm25.hpp -- C++ host mirror of the reference's interface for the query path, over the
// C ABI of vbm25.h (header only).
//
// The reference is Rust; there is no rustc in the build image, so the host side that a
// maintainer would write in Rust is mirrored here in C++ with the same
// names, argument meaning and error behaviour:
//   vbm25::intern        crates/bm25/src/vector.rs:19-35   (incl. the BLAKE3 keyed hash of long lexemes)
//   vbm25::Query         crates/bm25/src/vector.rs:96-134  (sorted unique 16-byte keys)
//   vbm25::Index::search crates/bm25/src/search.rs:28-36   (bm25::search, filter == true)
//   vbm25::search_growing, vbm25::merge_growing   crates/bm25/src/search.rs:83-135  (unsealed documents, host side)
//   vbm25::Segment::from_pages, vbm25::growing_from_pages   the relation's pages -> flat arrays (tape.rs, tuples.rs)
// Reference panics ("data corruption", "invalid data") and pgrx::error! become vbm25::Error.
#ifndef VBM25_HPP
#define VBM25_HPP

#include <algorithm>
#include <array>
#include <cstring>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

#include "vbm25.h"

namespace vbm25 {

constexpr size_t WIDTH = 16;  // crates/bm25/src/lib.rs:37
using Key = std::array<uint8_t, WIDTH>;
using Hit = vbm25_hit;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc) {
    if (rc != VBM25_OK) throw Error(rc, vbm25_last_error());
}

using Seed = std::array<uint8_t, 32>;  // MetaTuple.seed (tuples.rs:48-57)

// vector.rs:19-35: short lexemes are zero padded; lexemes of 16 bytes or more (or containing NUL) are the
// first 16 bytes of blake3::keyed_hash(seed, lexeme), last byte forced non-zero.
inline Key intern(const Seed &seed, std::string_view s) {
    Key k{};
    check(vbm25_intern(seed.data(), reinterpret_cast<const uint8_t *>(s.data()), s.size(), k.data()));
    return k;
}
// short path only (no index at hand): a lexeme that needs the hash raises VBM25_ERR_INVALID
inline Key intern(std::string_view s) {
    Key k{};
    check(vbm25_intern(nullptr, reinterpret_cast<const uint8_t *>(s.data()), s.size(), k.data()));
    return k;
}

// vector.rs:96-134
class Query {
  public:
    explicit Query(std::vector<Key> keys) : keys_(std::move(keys)) {
        for (size_t i = 1; i < keys_.size(); ++i)
            if (!(keys_[i - 1] < keys_[i])) throw Error(VBM25_ERR_INVALID, "invalid data");  // Query::new
    }
    // cast_tsvector_to_query, src/datatype/tsvector.rs:96-105: intern, sort, dedup
    template <class It>
    static Query from_tokens(It first, It last, const Seed *seed = nullptr) {
        std::vector<Key> keys;
        for (; first != last; ++first) keys.push_back(seed ? intern(*seed, *first) : intern(*first));
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        return Query(std::move(keys));
    }
    const std::vector<Key> &keys() const { return keys_; }
    size_t len() const { return keys_.size(); }
    bool is_empty() const { return keys_.empty(); }

  private:
    std::vector<Key> keys_;
};

// HBM-resident sealed segment.
class Index {
  public:
    Index(const vbm25_index_desc &desc, int device = 0) { check(vbm25_index_create(&desc, device, &h_)); }
    ~Index() { vbm25_index_destroy(h_); }
    Index(const Index &) = delete;
    Index &operator=(const Index &) = delete;

    // bm25::search(&index, k, &query, |_| true): best first, at most k hits.
    std::vector<Hit> search(size_t k, const Query &query) const {
        std::vector<uint32_t> ids(query.len());
        if (!ids.empty())
            check(vbm25_lookup_terms(h_, query.keys()[0].data(), uint32_t(ids.size()), ids.data()));
        // unknown tokens are ignored (search.rs:59-61); key order == term id order
        ids.erase(std::remove(ids.begin(), ids.end(), UINT32_MAX), ids.end());
        const uint32_t off[2] = {0, uint32_t(ids.size())};
        std::vector<Hit> hits(k);
        uint32_t n = 0;
        check(vbm25_search_batch(h_, ids.data(), off, 1, uint32_t(k), hits.data(), &n));
        hits.resize(n);
        return hits;
    }
    // The batched form the GPU is built for: queries as CSR of ascending term ids.
    void search_batch(const std::vector<uint32_t> &term_ids, const std::vector<uint32_t> &q_off, size_t k,
                      std::vector<Hit> &hits, std::vector<uint32_t> &n_hits) const {
        const uint32_t nq = uint32_t(q_off.size() - 1);
        hits.resize(size_t(nq) * k);
        n_hits.resize(nq);
        check(vbm25_search_batch(h_, term_ids.data(), q_off.data(), nq, uint32_t(k), hits.data(), n_hits.data()));
    }
    vbm25_index *handle() const { return h_; }

  private:
    vbm25_index *h_ = nullptr;
};

// Host copy of a flattened sealed segment (RAII over vbm25_segment).
class Segment {
  public:
    // A bm25 index relation in the reference's on-disk format (PostgreSQL 8 KiB pages): Meta -> Jump ->
    // documents / tokens / summaries / blocks tapes (the walk of maintain.rs:104-161).
    static Segment from_pages(vbm25_read_page_fn read_page, void *ctx) {
        vbm25_segment *h = nullptr;
        check(vbm25_segment_from_pages(read_page, ctx, &h));
        return Segment(h);
    }
    explicit Segment(vbm25_segment *h) : h_(h) {}
    Segment(Segment &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Segment(const Segment &) = delete;
    Segment &operator=(const Segment &) = delete;
    ~Segment() { vbm25_segment_free(h_); }
    vbm25_index_desc desc() const {
        vbm25_index_desc d;
        check(vbm25_segment_desc(h_, &d));
        return d;
    }

  private:
    vbm25_segment *h_;
};

// The growing (unsealed) segment, search.rs:83-135: documents as CSR over their VectorTuple elements.
struct GrowingDocs {
    std::vector<uint64_t> start{0};      // n + 1 offsets into key / tf
    std::vector<Key> key;                // Element.key
    std::vector<uint32_t> tf;            // Element.value
    std::vector<uint8_t> fieldnorm;      // VectorTuple::_2.fieldnorm
    std::vector<uint16_t> payload;       // VectorTuple::_0.payload, n x 3
    std::vector<uint8_t> deleted;        // VectorTuple::_0.deleted
    size_t size() const { return start.size() - 1; }
};
// The unsealed documents of a relation (vectors tape from Jump.ptr_vectors; VectorTuple _2 / _1 / _0).
inline GrowingDocs growing_from_pages(vbm25_read_page_fn read_page, void *ctx) {
    vbm25_growing *g = nullptr;
    check(vbm25_growing_from_pages(read_page, ctx, &g));
    vbm25_growing_desc d;
    const int rc = vbm25_growing_get_desc(g, &d);
    GrowingDocs out;
    if (rc == VBM25_OK) {
        out.start.assign(d.start, d.start + d.n_docs + 1);
        out.key.resize(d.n_elements);
        if (d.n_elements) std::memcpy(out.key[0].data(), d.key, 16 * d.n_elements);
        out.tf.assign(d.tf, d.tf + d.n_elements);
        out.fieldnorm.assign(d.fieldnorm, d.fieldnorm + d.n_docs);
        out.payload.assign(d.payload, d.payload + 3ull * d.n_docs);
        out.deleted.assign(d.deleted, d.deleted + d.n_docs);
    }
    vbm25_growing_free(g);
    check(rc);
    return out;
}
// Scores the unsealed documents with the sealed segment's statistics (host code, as in the reference).
inline std::vector<Hit> search_growing(const vbm25_index_desc &desc, const Query &query, size_t k,
                                       const GrowingDocs &g) {
    std::vector<Hit> hits(k);
    uint32_t n = 0;
    check(vbm25_growing_search(&desc, query.is_empty() ? nullptr : query.keys()[0].data(), uint32_t(query.len()),
                               uint32_t(k), uint32_t(g.size()), g.start.data(),
                               g.key.empty() ? nullptr : g.key[0].data(), g.tf.data(), g.fieldnorm.data(),
                               g.payload.data(), g.deleted.empty() ? nullptr : g.deleted.data(), hits.data(), &n));
    hits.resize(n);
    return hits;
}
// search.rs:83-135 + 301-313: top-k of the union of the sealed hits (device) and the growing hits
// (host), best first; on equal scores sealed hits come first.
inline std::vector<Hit> merge_growing(const std::vector<Hit> &sealed, const std::vector<Hit> &grow, size_t k) {
    std::vector<Hit> out(k);
    uint32_t n = 0;
    check(vbm25_merge_hits(sealed.data(), uint32_t(sealed.size()), grow.data(), uint32_t(grow.size()), uint32_t(k),
                           out.data(), &n));
    out.resize(n);
    return out;
}

}  // namespace vbm25
#endif
