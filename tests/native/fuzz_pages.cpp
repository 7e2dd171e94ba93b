// AddressSanitizer harness for the host-side page reader (vectorchord-bm25_amd/csrc/pages.cpp):
// a relation written by oracle/pages.cpp, damaged at random, must be flattened or rejected without
// any out-of-bounds access.  Built and run by tests/test_pages.py::test_page_reader_under_asan.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/vbm25.h"
#include "../../oracle/oracle.h"

namespace vbm25 {
int set_error(int code, const char *, ...) { return code; }  // the library defines it in search.hip
}

struct Rel {
    std::vector<std::vector<uint8_t>> pages;
};
static const uint8_t *read_page(void *ctx, uint32_t id) {
    auto *r = static_cast<Rel *>(ctx);
    return id < r->pages.size() ? r->pages[id].data() : nullptr;
}

int main() {
    // a small corpus: 40 terms over 3000 documents
    std::mt19937_64 rng(7);
    const uint32_t n_docs = 3000, n_terms = 40;
    std::vector<uint32_t> doc_len(n_docs, 0), post_doc, post_tf;
    std::vector<uint16_t> payload(3 * n_docs, 1);
    std::vector<uint64_t> term_start{0};
    std::vector<uint8_t> keys(16 * n_terms, 0);
    for (uint32_t t = 0; t < n_terms; ++t) {
        std::snprintf(reinterpret_cast<char *>(&keys[16 * t]), 16, "k%03u", t);
        for (uint32_t d = 0; d < n_docs; ++d)
            if (rng() % 7 == 0) {
                post_doc.push_back(d);
                post_tf.push_back(1 + rng() % 4);
                doc_len[d] += post_tf.back();
            }
        term_start.push_back(post_doc.size());
    }
    for (auto &l : doc_len) l = l ? l : 1;
    orc_index *ix = orc_index_build(1.2, 0.75, n_docs, doc_len.data(), payload.data(), n_terms, keys.data(),
                                    term_start.data(), post_doc.data(), post_tf.data());
    orc_pages *op = orc_pages_build(ix, nullptr);
    for (int i = 0; i < 30; ++i) {
        const uint16_t pl[3] = {uint16_t(i), 2, 3};
        std::vector<uint32_t> tfs(1 + rng() % 600, 2);
        std::vector<uint8_t> k(16 * tfs.size(), 0);
        for (size_t j = 0; j < tfs.size(); ++j) std::snprintf(reinterpret_cast<char *>(&k[16 * j]), 16, "g%05zu", j);
        orc_pages_insert(op, pl, uint32_t(tfs.size()), k.data(), tfs.data());
    }
    Rel clean;
    for (uint32_t i = 0; i < orc_pages_count(op); ++i) clean.pages.emplace_back(orc_pages_get(op, i), orc_pages_get(op, i) + 8192);
    int ok = 0, bad = 0;
    for (int it = 0; it < 4000; ++it) {
        Rel r = clean;
        if (it) {
            const int flips = 1 + rng() % 4;
            for (int f = 0; f < flips; ++f) {
                auto &pg = r.pages[rng() % r.pages.size()];
                const uint32_t pos = (rng() % 3 == 0) ? 12 + rng() % 60 : (rng() % 3 == 0 ? 8184 + rng() % 8 : rng() % 8192);
                pg[pos] = uint8_t(rng());
            }
        }
        vbm25_segment *seg = nullptr;
        if (vbm25_segment_from_pages(read_page, &r, &seg) == VBM25_OK) {
            vbm25_index_desc d;
            vbm25_segment_desc(seg, &d);
            vbm25_segment_free(seg);
            ++ok;
        } else {
            ++bad;
        }
        vbm25_growing *g = nullptr;
        if (vbm25_growing_from_pages(read_page, &r, &g) == VBM25_OK) {
            vbm25_growing_desc gd;
            vbm25_growing_get_desc(g, &gd);
            vbm25_growing_free(g);
        }
    }
    std::printf("fuzz done: %d flattened, %d rejected\n", ok, bad);
    orc_pages_free(op);
    orc_index_free(ix);
    return ok > 0 && bad > 0 ? 0 : 1;
}
