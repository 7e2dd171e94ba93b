// decode.h -- block decode (compression.rs:65-136 formats): one wave, two postings per lane.
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, block_fetch, topk_reg, scan_range, scan_dense, scan_many, merge.

// ---------------------------------------------------------------------------
// Block decode: one wave, two postings per lane (value indices 2*lane, 2*lane+1)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bp_field(const uint32_t *__restrict__ w32, uint32_t b,
                                             uint32_t i) {
    // crates/simd/src/bitpacking.rs:58-98: lane l = i % 4 is an LSB-first stream of b-bit
    // fields, its word w lives at 32-bit index 4*w + l.
    const uint32_t l = i & 3, bit = (i >> 2) * b, w = bit >> 5, sh = bit & 31;
    const uint32_t lo = w32[4 * w + l];
    const uint32_t hi = (sh + b > 32) ? w32[4 * (w + 1) + l] : 0u;
    const unsigned long long both = ((unsigned long long)hi << 32) | lo;
    return (uint32_t)(both >> sh) & ((1u << b) - 1u);
}

__device__ __forceinline__ uint32_t byte_field(const uint8_t *__restrict__ p, uint32_t w,
                                               uint32_t i) {
    uint32_t v = 0;
    for (uint32_t j = 0; j < w; ++j) v |= (uint32_t)p[i * w + j] << (8 * j);
    return v;
}

// Raw fields of a block payload (no delta).  meta: bit 7 = byte packed, low bits = width.
__device__ __forceinline__ void decode_fields(const uint8_t *__restrict__ p, uint32_t meta,
                                              uint32_t n, uint32_t lane, uint32_t &v0,
                                              uint32_t &v1) {
    const uint32_t i0 = 2 * lane, i1 = i0 + 1;
    const uint32_t width = meta & 127u;
    v0 = 0;
    v1 = 0;
    if ((meta >> 7) == 0) {
        const uint32_t *w32 = reinterpret_cast<const uint32_t *>(p);
        if (width == 32) {
            v0 = w32[i0];
            v1 = w32[i1];
        } else if (width != 0) {
            v0 = bp_field(w32, width, i0);
            v1 = bp_field(w32, width, i1);
        }
    } else {
        if (i0 < n) v0 = byte_field(p, width, i0);
        if (i1 < n) v1 = byte_field(p, width, i1);
    }
}

__device__ __forceinline__ uint32_t payload_bytes(uint32_t meta, uint32_t n) {
    return (meta >> 7) ? (meta & 127u) * n : 16u * (meta & 127u);
}

// Document ids of a block: d1 deltas in index order from min_doc
// (bitpacking_u32_ordered.rs:191-218), except width 32 / bytewidth 4 = raw absolute.
__device__ __forceinline__ void decode_doc_ids(const uint8_t *__restrict__ p, uint32_t meta,
                                               uint32_t n, uint32_t min_doc, uint32_t lane,
                                               uint32_t &d0, uint32_t &d1) {
    uint32_t v0, v1;
    decode_fields(p, meta, n, lane, v0, v1);
    const uint32_t width = meta & 127u;
    const bool raw = (meta >> 7) ? (width == 4) : (width == 32);
    if (raw) {
        d0 = v0;
        d1 = v1;
        return;
    }
    // inclusive scan of the per-lane sums over the wave: DPP row shifts and row broadcasts (round 6: six __shfl_up were six dependent
    // round trips through the LDS crossbar per decoded block -- the in-kernel decode of an index without post_rel16, the cold pass
    // of scan_win_kernel and the slow tasks of scan_dense_kernel all come through here).  All 64 lanes are active at every call site.
    const uint32_t own = v0 + v1;
    uint32_t x = own;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31
    d0 = min_doc + (x - own) + v0;
    d1 = d0 + v1;
}
