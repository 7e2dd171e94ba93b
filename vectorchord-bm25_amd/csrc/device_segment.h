// device_segment.h -- a sealed segment whose arrays live in HBM (vbm25_device_segment): what csrc/flush.hip builds and
// csrc/search.hip (vbm25_index_create_from_device) makes an index of without a round trip through the host.
// Shared by the two HIP translation units of libvbm25; not part of the ABI.
#ifndef VBM25_DEVICE_SEGMENT_H
#define VBM25_DEVICE_SEGMENT_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace vbm25 {

struct HbmArray {
    void *p = nullptr;
    size_t bytes = 0;
    HbmArray() = default;
    HbmArray(const HbmArray &) = delete;
    HbmArray &operator=(const HbmArray &) = delete;
    ~HbmArray() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) {
        bytes = n;
        return hipMalloc(&p, n ? n : 16);
    }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

}  // namespace vbm25

// The values flush.rs:40-158 writes to the Token / Summary / Block / Document tapes (the arrays of vbm25_index_desc), in the
// HBM of `device`; the few things the host needs again (keys for the token lookup, block counts) as host copies.
struct vbm25_device_segment {
    int device = 0;
    double k1 = 1.2, b = 0.75;
    uint32_t n_docs = 0, n_terms = 0, n_blocks = 0;
    uint64_t sum_len = 0, blob_bytes = 0;
    std::vector<uint8_t> term_key;           // n_terms x 16, ascending
    std::vector<uint32_t> term_first_block;  // n_terms + 1
    std::vector<uint32_t> term_df;           // n_terms
    std::vector<uint64_t> term_bytes;        // per term: the algorithmic bytes of its postings (vbm25_query_bytes without the 14 k)
    std::vector<uint32_t> token_term;        // synthetic corpora only: token number -> term id
    vbm25::HbmArray d_term_df, d_term_wand_fn, d_term_wand_tf, d_term_first_block, d_blk_min, d_blk_max, d_blk_n, d_blk_wand_fn,
        d_blk_wand_tf, d_blk_meta_doc, d_blk_meta_tf, d_blk_off8, d_blob, d_doc_fieldnorm, d_doc_payload;
};

#endif
