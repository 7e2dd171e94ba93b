// vbm25_internal.h -- shared between the host (segment.cpp) and device (search.hip) halves
// of libvbm25.  Not part of the ABI.
#ifndef VBM25_INTERNAL_H
#define VBM25_INTERNAL_H

#include <cstdint>
#include <exception>
#include <new>
#include <vector>

#include "../../include/vbm25.h"

namespace vbm25 {

// Thread-local error text; returns `code` so callers can `return set_error(...)`.
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// No C++ exception may cross the C ABI (the reference's crate is #![deny(ffi_unwind_calls)], src/lib.rs:16):
// every entry point that allocates runs its body through this.
template <class F>
int guarded(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return set_error(VBM25_ERR_NOMEM, "out of host memory");
    } catch (const std::exception &e) {
        return set_error(VBM25_ERR_INVALID, "internal error: %s", e.what());
    } catch (...) {
        return set_error(VBM25_ERR_INVALID, "internal error");
    }
}

// bm25.rs:15-283
const uint32_t *fieldnorm_lengths();
uint8_t length_to_fieldnorm(uint32_t length);
// Cache::new (bm25.rs:340-354): the per-index s1[256] table and the per-term s0
void bm25_tables(uint32_t n_docs, uint64_t sum_len, double k1, double b, double *s1_256);
double bm25_s0(uint32_t n_docs, uint32_t df, double k1);

int check_desc(const vbm25_index_desc *d);
// BLAKE3 hash (key32 == NULL) or keyed hash of `in`; out_len bytes of output
void blake3(const uint8_t *key32, const uint8_t *in, size_t len, uint8_t *out, size_t out_len);

// Host copy of a flattened sealed segment (the arrays of vbm25_index_desc).
struct Segment {
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
    std::vector<uint32_t> token_term;  // synthetic corpora only: token number -> term id
    void desc(vbm25_index_desc *d) const;
};

}  // namespace vbm25

struct vbm25_segment : vbm25::Segment {};

#endif
