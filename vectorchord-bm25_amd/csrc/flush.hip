// flush.hip -- sealed-segment construction on the device: the encode side of the posting
// blocks, /root/reference/crates/bm25/src/flush.rs:40-158 with compression.rs:36-63,94-110 (bit width = OR
// of the deltas, 4-lane vertical packing, byte-packed tails) and the WAND pairs of bm25.rs:297-332 (first
// maximiser under strict '<', per block and per token).  Output: the same vbm25_segment the host builder
// (segment.cpp) makes, byte for byte -- tests/test_gpu_flush.py compares every array.
//
// One wave per 128-posting block, two postings per lane:
//   block_stats_kernel  deltas (DPP neighbour), OR-reduce -> widths, body length, min / max, block WAND pair
//                       (f64 tf() of bm25.rs:291-295 per posting, wave arg-max that keeps the FIRST maximum),
//                       input validation (ids < n_docs, strictly increasing inside a term, tf > 0)
//   (exclusive scan of the body lengths -> blk_off8, hipcub)
//   block_pack_kernel   fields OR-ed into an LDS image of the body (a field may straddle two words of its lane
//                       stream), copied out coalesced; tails and width-32 blocks byte / word copies
//   term_wand_kernel    Wand::extend over a token's blocks in order (first maximum again)
//   doc_kernel          fieldnorm code of every document (length_to_fieldnorm) and the sum of lengths
// Host memory in, host memory out; the HBM copies live for the duration of the call.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "vbm25_internal.h"
#include "device_segment.h"

namespace {

using namespace vbm25;

#define FL_TRY(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return set_error(VBM25_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DBuf {
    void *p = nullptr;
    ~DBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 16); }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

struct WidenU32 {
    __host__ __device__ unsigned long long operator()(uint32_t v) const { return v; }
};

struct FlushArgs {
    uint32_t n_docs, n_terms, n_blocks;
    const uint64_t *term_start;         // n_terms + 1
    const uint32_t *term_first_block;   // n_terms + 1
    const uint32_t *post_doc, *post_tf;
    const uint8_t *fieldnorm;           // per document
    const double *denom;                // 256: k1 * (1 - b + b * len(f) / avgdl)
    double kp1;
    // per block
    uint32_t *min_doc, *max_doc, *wand_tf, *len8, *off8;
    uint8_t *n, *wand_fn, *meta_doc, *meta_tf;
    double *wand_val;
    uint8_t *blob;
    uint32_t *error_flag;
};

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t width_of_dev(uint32_t ored) { return ored ? 32u - (uint32_t)__clz(ored) : 0u; }

// term of block j: term_first_block[t] <= j < term_first_block[t + 1]
__device__ __forceinline__ uint32_t term_of_block(const uint32_t *tfb, uint32_t n_terms, uint32_t j) {
    uint32_t lo = 0, hi = n_terms;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tfb[mid] <= j) lo = mid; else hi = mid;
    }
    return lo;
}

// the block's postings, two per lane (indices 2 lane, 2 lane + 1), and the id before the block's first one
struct BlockIn {
    uint32_t n, d0, d1, t0, t1, delta0, delta1, prev;  // prev: the id before d0 (lane - 1's second id)
    bool has0, has1;
};
__device__ __forceinline__ BlockIn load_block(const FlushArgs &a, uint32_t j, uint32_t lane, uint32_t &term) {
    term = term_of_block(a.term_first_block, a.n_terms, j);
    const uint64_t p0 = a.term_start[term] + 128ull * (j - a.term_first_block[term]);
    const uint64_t pe = a.term_start[term + 1];
    BlockIn b;
    b.n = (uint32_t)min((uint64_t)128, pe - p0);
    b.has0 = 2 * lane < b.n;
    b.has1 = 2 * lane + 1 < b.n;
    b.d0 = b.has0 ? a.post_doc[p0 + 2 * lane] : 0u;
    b.d1 = b.has1 ? a.post_doc[p0 + 2 * lane + 1] : 0u;
    b.t0 = b.has0 ? a.post_tf[p0 + 2 * lane] : 0u;
    b.t1 = b.has1 ? a.post_tf[p0 + 2 * lane + 1] : 0u;
    b.prev = __shfl_up(b.d1, 1);
    b.delta0 = b.has0 ? (lane == 0 ? 0u : b.d0 - b.prev) : 0u;  // the first delta is against the block's own first id
    b.delta1 = b.has1 ? b.d1 - b.d0 : 0u;
    return b;
}

__global__ void __launch_bounds__(256) block_stats_kernel(FlushArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t j = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (j >= a.n_blocks) return;
    uint32_t term;
    const BlockIn b = load_block(a, j, lane, term);
    // validation (segment.rs:19-50 invariants the host builder checks too)
    bool bad = false;
    if (b.has0) bad |= b.d0 >= a.n_docs || b.t0 == 0 || (lane > 0 && b.d0 <= b.prev);
    if (b.has1) bad |= b.d1 >= a.n_docs || b.t1 == 0 || b.d1 <= b.d0;
    if (lane == 0 && j > a.term_first_block[term]) {  // across blocks of a term
        const uint64_t p0 = a.term_start[term] + 128ull * (j - a.term_first_block[term]);
        bad |= a.post_doc[p0 - 1] >= b.d0;
    }
    if (bad) atomicOr(a.error_flag, 1u);
    const uint32_t bd = width_of_dev(wave_or_u32(b.delta0 | b.delta1)), bt = width_of_dev(wave_or_u32(b.t0 | b.t1));
    uint32_t meta_d, meta_t, len_d, len_t;
    if (b.n == 128) {
        meta_d = bd;
        meta_t = bt;
        len_d = 16 * bd;
        len_t = 16 * bt;
    } else {
        const uint32_t wd = max(1u, (bd + 7) / 8), wt = max(1u, (bt + 7) / 8);
        meta_d = 0x80u | wd;
        meta_t = 0x80u | wt;
        len_d = wd * b.n;
        len_t = wt * b.n;
    }
    // block WAND pair: first maximiser, strict '<' (bm25.rs:311-318)
    double v = 0.0;
    uint32_t vi = 0xffffffffu, vfn = 255, vtf = 0;
    if (b.has0) {
        const uint32_t f = a.fieldnorm[min(b.d0, a.n_docs - 1)];
        const double t = (double)b.t0, x = (t * a.kp1) / (t + a.denom[f]);
        if (v < x) {
            v = x;
            vi = 2 * lane;
            vfn = f;
            vtf = b.t0;
        }
    }
    if (b.has1) {
        const uint32_t f = a.fieldnorm[min(b.d1, a.n_docs - 1)];
        const double t = (double)b.t1, x = (t * a.kp1) / (t + a.denom[f]);
        if (v < x) {
            v = x;
            vi = 2 * lane + 1;
            vfn = f;
            vtf = b.t1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const uint32_t oi = __shfl_xor(vi, o), ofn = __shfl_xor(vfn, o), otf = __shfl_xor(vtf, o);
        if (ov > v || (ov == v && oi < vi)) {
            v = ov;
            vi = oi;
            vfn = ofn;
            vtf = otf;
        }
    }
    if (lane == 0) {
        a.min_doc[j] = b.d0;
        a.n[j] = (uint8_t)b.n;
        a.wand_fn[j] = (uint8_t)vfn;
        a.wand_tf[j] = vtf;
        a.wand_val[j] = v;
        a.meta_doc[j] = (uint8_t)meta_d;
        a.meta_tf[j] = (uint8_t)meta_t;
        a.len8[j] = (((len_d + 7) & ~7u) + ((len_t + 7) & ~7u)) / 8;
    }
    const uint32_t last = b.n - 1;
    if (lane == last / 2) a.max_doc[j] = (last & 1) ? b.d1 : b.d0;
}

// value index i = 2 lane + e: lane stream i % 4, step i / 4 (crates/simd/src/bitpacking.rs:58-98)
__device__ __forceinline__ void or_field(uint32_t *img, uint32_t i, uint32_t v, uint32_t b) {
    const uint32_t l = i & 3, bit = (i >> 2) * b, w = bit >> 5, sh = bit & 31;
    atomicOr(&img[4 * w + l], v << sh);
    if (sh + b > 32) atomicOr(&img[4 * (w + 1) + l], v >> (32 - sh));
}

__global__ void __launch_bounds__(256) block_pack_kernel(FlushArgs a) {
    __shared__ uint32_t s_img[4][2][128];  // per wave: doc-id body and tf body, up to 512 bytes each
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t j = blockIdx.x * (blockDim.x / 64) + wv;
    if (j >= a.n_blocks) return;
    uint32_t term;
    const BlockIn b = load_block(a, j, lane, term);
    const uint32_t md = a.meta_doc[j], mt = a.meta_tf[j];
    uint32_t *imd = s_img[wv][0], *imt = s_img[wv][1];
    imd[lane] = imd[lane + 64] = imt[lane] = imt[lane + 64] = 0;
    __builtin_amdgcn_wave_barrier();
    uint32_t len_d, len_t;
    if (b.n == 128) {
        len_d = 16 * md;
        len_t = 16 * mt;
        if (md == 32) {  // raw absolute ids
            imd[2 * lane] = b.d0;
            imd[2 * lane + 1] = b.d1;
        } else if (md) {
            or_field(imd, 2 * lane, b.delta0, md);
            or_field(imd, 2 * lane + 1, b.delta1, md);
        }
        if (mt == 32) {
            imt[2 * lane] = b.t0;
            imt[2 * lane + 1] = b.t1;
        } else if (mt) {
            or_field(imt, 2 * lane, b.t0, mt);
            or_field(imt, 2 * lane + 1, b.t1, mt);
        }
    } else {  // tail: byte-packed (compression.rs:53-62), width-4 document ids are absolute
        const uint32_t wd = md & 127u, wt = mt & 127u;
        len_d = wd * b.n;
        len_t = wt * b.n;
        uint8_t *bd8 = reinterpret_cast<uint8_t *>(imd), *bt8 = reinterpret_cast<uint8_t *>(imt);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t i = 2 * lane + e;
            if (i < b.n) {
                const uint32_t v = wd == 4 ? (e ? b.d1 : b.d0) : (e ? b.delta1 : b.delta0);
                const uint32_t t = e ? b.t1 : b.t0;
                for (uint32_t q = 0; q < wd; ++q) bd8[i * wd + q] = (uint8_t)(v >> (8 * q));
                for (uint32_t q = 0; q < wt; ++q) bt8[i * wt + q] = (uint8_t)(t >> (8 * q));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // copy out: doc bytes, zero pad to 8, tf bytes, zero pad to 8 (the image is zero beyond the payload)
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.blob + 8ull * a.off8[j]);
    const uint32_t wd4 = ((len_d + 7) & ~7u) / 4, wt4 = ((len_t + 7) & ~7u) / 4;
    for (uint32_t i = lane; i < wd4; i += 64) dst[i] = imd[i];
    for (uint32_t i = lane; i < wt4; i += 64) dst[wd4 + i] = imt[i];
}

__global__ void __launch_bounds__(64) term_wand_kernel(FlushArgs a, uint8_t *term_wand_fn, uint32_t *term_wand_tf, uint32_t *term_df) {
    const uint32_t t = blockIdx.x, lane = threadIdx.x;
    const uint32_t j0 = a.term_first_block[t], j1 = a.term_first_block[t + 1];
    double v = 0.0;
    uint32_t vi = 0xffffffffu;
    for (uint32_t j = j0 + lane; j < j1; j += 64) {  // Wand::extend, bm25.rs:319-325: strict '<' keeps the first
        const double x = a.wand_val[j];
        if (v < x) {
            v = x;
            vi = j;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const uint32_t oi = __shfl_xor(vi, o);
        if (ov > v || (ov == v && oi < vi)) {
            v = ov;
            vi = oi;
        }
    }
    if (lane == 0) {
        term_wand_fn[t] = vi == 0xffffffffu ? 255 : a.wand_fn[vi];
        term_wand_tf[t] = vi == 0xffffffffu ? 0 : a.wand_tf[vi];
        term_df[t] = (uint32_t)(a.term_start[t + 1] - a.term_start[t]);
    }
}

__global__ void __launch_bounds__(256) doc_kernel(uint32_t n_docs, const uint32_t *doc_len, const uint32_t *fn_len, uint8_t *fieldnorm,
                                                  unsigned long long *sum_len) {
    unsigned long long local = 0;
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += gridDim.x * blockDim.x) {
        const uint32_t len = doc_len[d];
        uint32_t lo = 0, hi = 256;  // length_to_fieldnorm, bm25.rs:278-283
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (fn_len[mid] <= len) lo = mid; else hi = mid;
        }
        fieldnorm[d] = (uint8_t)lo;
        local += len;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum_len, local);
}

__global__ void __launch_bounds__(256) gather_u32_kernel(uint32_t n, const uint32_t *idx, const uint32_t *src, uint32_t *dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
// synthetic ctids (fetcher.rs:218-225's layout: 64 tuples per heap block)
__global__ void __launch_bounds__(256) synth_payload_kernel(uint32_t n_docs, uint16_t *payload) {
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += gridDim.x * blockDim.x) {
        const uint32_t blk = d / 64;
        payload[3ull * d + 0] = (uint16_t)(blk >> 16);
        payload[3ull * d + 1] = (uint16_t)(blk & 0xffff);
        payload[3ull * d + 2] = (uint16_t)(d % 64 + 1);
    }
}

// The encode, everything left in HBM (vbm25_device_segment).  Inputs on the host, or already on the device: dev_len (document
// lengths), dev_doc / dev_tf (the mappings, sorted by (token, document)); doc_payload == nullptr: synthetic ctids.
int build_device_core(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len, const uint32_t *dev_len,
                      const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key, const uint64_t *term_start,
                      const uint32_t *post_doc, const uint32_t *post_tf, const uint32_t *dev_doc, const uint32_t *dev_tf,
                      std::unique_ptr<vbm25_device_segment> &out) {
    if ((!doc_len && !dev_len) || !term_start || (n_terms && (!term_key || ((!post_doc || !post_tf) && (!dev_doc || !dev_tf)))))
        return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (!n_docs) return set_error(VBM25_ERR_INVALID, "segment without documents");
    if (!(k1 >= 1.2 && k1 <= 2.0) || !(b >= 0.0 && b <= 1.0))  // types.rs:18-45
        return set_error(VBM25_ERR_INVALID, "k1 must be in [1.2, 2] and b in [0, 1]");
    for (uint32_t t = 0; t + 1 < n_terms; ++t)
        if (std::memcmp(term_key + 16ull * t, term_key + 16ull * (t + 1), 16) >= 0)
            return set_error(VBM25_ERR_INVALID, "term keys must be strictly ascending");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return set_error(VBM25_ERR_DEVICE, "no HIP device: the device builder has no CPU fallback (vbm25_segment_build is the host builder)");
    if (device < 0 || device >= n_dev) return set_error(VBM25_ERR_INVALID, "device %d out of range (%d devices)", device, n_dev);
    FL_TRY(hipSetDevice(device));

    auto ds = std::make_unique<vbm25_device_segment>();
    ds->device = device;
    ds->k1 = k1;
    ds->b = b;
    ds->n_docs = n_docs;
    ds->n_terms = n_terms;
    ds->term_key.assign(term_key, term_key + 16ull * n_terms);
    ds->term_first_block.resize(size_t(n_terms) + 1);
    ds->term_df.resize(n_terms);
    uint64_t nb = 0;
    for (uint32_t t = 0; t < n_terms; ++t) {
        if (term_start[t + 1] <= term_start[t]) return set_error(VBM25_ERR_INVALID, "term %u has no postings", t);
        if (term_start[t + 1] - term_start[t] > 0xffffffffull) return set_error(VBM25_ERR_UNSUPPORTED, "a token with more than 2^32 mappings");
        ds->term_first_block[t] = uint32_t(nb);
        ds->term_df[t] = uint32_t(term_start[t + 1] - term_start[t]);
        nb += (term_start[t + 1] - term_start[t] + 127) / 128;
        if (nb > 0xfffffff0ull) return set_error(VBM25_ERR_UNSUPPORTED, "more than 2^32 blocks");
    }
    ds->term_first_block[n_terms] = uint32_t(nb);
    const uint32_t n_blocks = ds->n_blocks = uint32_t(nb);
    const uint64_t n_post = n_terms ? term_start[n_terms] : 0;

    DBuf d_len, d_fnlen, d_sum, d_ts, d_pd, d_pt, d_denom, d_err, b_len8, b_wval, d_tmp, d_bnd;
    FL_TRY(d_fnlen.alloc(4 * 256));
    FL_TRY(ds->d_doc_fieldnorm.alloc(n_docs));
    FL_TRY(ds->d_doc_payload.alloc(6ull * n_docs));
    FL_TRY(d_sum.alloc(8));
    FL_TRY(d_err.alloc(4));
    if (!dev_len) {
        FL_TRY(d_len.alloc(4ull * n_docs));
        FL_TRY(hipMemcpy(d_len.p, doc_len, 4ull * n_docs, hipMemcpyHostToDevice));
        dev_len = d_len.as<uint32_t>();
    }
    if (doc_payload) FL_TRY(hipMemcpy(ds->d_doc_payload.p, doc_payload, 6ull * n_docs, hipMemcpyHostToDevice));
    else synth_payload_kernel<<<1024, 256>>>(n_docs, ds->d_doc_payload.as<uint16_t>());
    FL_TRY(hipMemcpy(d_fnlen.p, fieldnorm_lengths(), 4 * 256, hipMemcpyHostToDevice));
    FL_TRY(hipMemset(d_sum.p, 0, 8));
    FL_TRY(hipMemset(d_err.p, 0, 4));
    doc_kernel<<<1024, 256>>>(n_docs, dev_len, d_fnlen.as<uint32_t>(), ds->d_doc_fieldnorm.as<uint8_t>(), d_sum.as<unsigned long long>());
    FL_TRY(hipGetLastError());
    unsigned long long sum_len = 0;
    FL_TRY(hipMemcpy(&sum_len, d_sum.p, 8, hipMemcpyDeviceToHost));
    ds->sum_len = sum_len;
    double denom[256];
    bm25_tables(n_docs, sum_len, k1, b, denom);

    FL_TRY(d_ts.alloc(8ull * (n_terms + 1)));
    FL_TRY(ds->d_term_first_block.alloc(4ull * (n_terms + 1)));
    if (!dev_doc) {
        FL_TRY(d_pd.alloc(4ull * n_post));
        FL_TRY(d_pt.alloc(4ull * n_post));
    }
    FL_TRY(d_denom.alloc(8 * 256));
    FL_TRY(hipMemcpy(d_ts.p, term_start, 8ull * (n_terms + 1), hipMemcpyHostToDevice));
    FL_TRY(hipMemcpy(ds->d_term_first_block.p, ds->term_first_block.data(), 4ull * (n_terms + 1), hipMemcpyHostToDevice));
    if (n_post && !dev_doc) {
        FL_TRY(hipMemcpy(d_pd.p, post_doc, 4ull * n_post, hipMemcpyHostToDevice));
        FL_TRY(hipMemcpy(d_pt.p, post_tf, 4ull * n_post, hipMemcpyHostToDevice));
    }
    FL_TRY(hipMemcpy(d_denom.p, denom, 8 * 256, hipMemcpyHostToDevice));
    FL_TRY(ds->d_blk_min.alloc(4ull * n_blocks));
    FL_TRY(ds->d_blk_max.alloc(4ull * n_blocks));
    FL_TRY(ds->d_blk_wand_tf.alloc(4ull * n_blocks));
    FL_TRY(b_len8.alloc(4ull * (n_blocks + 1)));
    FL_TRY(ds->d_blk_off8.alloc(4ull * (n_blocks + 1)));
    FL_TRY(ds->d_blk_n.alloc(n_blocks));
    FL_TRY(ds->d_blk_wand_fn.alloc(n_blocks));
    FL_TRY(ds->d_blk_meta_doc.alloc(n_blocks));
    FL_TRY(ds->d_blk_meta_tf.alloc(n_blocks));
    FL_TRY(b_wval.alloc(8ull * n_blocks));
    FL_TRY(ds->d_term_wand_fn.alloc(n_terms));
    FL_TRY(ds->d_term_wand_tf.alloc(4ull * n_terms));
    FL_TRY(ds->d_term_df.alloc(4ull * n_terms));
    FlushArgs a{};
    a.n_docs = n_docs;
    a.n_terms = n_terms;
    a.n_blocks = n_blocks;
    a.term_start = d_ts.as<uint64_t>();
    a.term_first_block = ds->d_term_first_block.as<uint32_t>();
    a.post_doc = dev_doc ? dev_doc : d_pd.as<uint32_t>();
    a.post_tf = dev_tf ? dev_tf : d_pt.as<uint32_t>();
    a.fieldnorm = ds->d_doc_fieldnorm.as<uint8_t>();
    a.denom = d_denom.as<double>();
    a.kp1 = k1 + 1.0;
    a.min_doc = ds->d_blk_min.as<uint32_t>();
    a.max_doc = ds->d_blk_max.as<uint32_t>();
    a.wand_tf = ds->d_blk_wand_tf.as<uint32_t>();
    a.len8 = b_len8.as<uint32_t>();
    a.off8 = ds->d_blk_off8.as<uint32_t>();
    a.n = ds->d_blk_n.as<uint8_t>();
    a.wand_fn = ds->d_blk_wand_fn.as<uint8_t>();
    a.meta_doc = ds->d_blk_meta_doc.as<uint8_t>();
    a.meta_tf = ds->d_blk_meta_tf.as<uint8_t>();
    a.wand_val = b_wval.as<double>();
    a.error_flag = d_err.as<uint32_t>();
    ds->term_bytes.assign(n_terms, 0);
    if (n_blocks) {
        FL_TRY(hipMemset(b_len8.p, 0, 4ull * (n_blocks + 1)));
        block_stats_kernel<<<(n_blocks + 3) / 4, 256>>>(a);
        FL_TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        FL_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, a.len8, a.off8, (int)(n_blocks + 1)));
        FL_TRY(d_tmp.alloc(tmp_bytes));
        FL_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tmp_bytes, a.len8, a.off8, (int)(n_blocks + 1)));
        uint32_t flag = 0;
        FL_TRY(hipMemcpy(&flag, d_err.p, 4, hipMemcpyDeviceToHost));
        if (flag) return set_error(VBM25_ERR_INVALID, "mappings must be sorted by (token, document), ids < n_docs, tf > 0");
        // total body length: a 32-bit scan would wrap silently beyond 2^32 units of 8 bytes -- summed in 64 bits as well
        unsigned long long total8 = 0;
        {
            size_t rb = 0;
            hipcub::TransformInputIterator<unsigned long long, WidenU32, const uint32_t *> wide(a.len8, WidenU32());
            FL_TRY(hipcub::DeviceReduce::Sum(nullptr, rb, wide, d_sum.as<unsigned long long>(), (int)n_blocks));
            DBuf d_rt;
            FL_TRY(d_rt.alloc(rb));
            FL_TRY(hipcub::DeviceReduce::Sum(d_rt.p, rb, wide, d_sum.as<unsigned long long>(), (int)n_blocks));
            FL_TRY(hipMemcpy(&total8, d_sum.p, 8, hipMemcpyDeviceToHost));
        }
        if (total8 > 0xffffffffull) return set_error(VBM25_ERR_UNSUPPORTED, "block bodies exceed 32 GiB");
        ds->blob_bytes = 8ull * total8;
        FL_TRY(ds->d_blob.alloc(ds->blob_bytes));
        a.blob = ds->d_blob.as<uint8_t>();
        block_pack_kernel<<<(n_blocks + 3) / 4, 256>>>(a);
        FL_TRY(hipGetLastError());
        term_wand_kernel<<<n_terms, 64>>>(a, ds->d_term_wand_fn.as<uint8_t>(), ds->d_term_wand_tf.as<uint32_t>(), ds->d_term_df.as<uint32_t>());
        FL_TRY(hipGetLastError());
        // the body offsets at the tokens' first blocks: the algorithmic bytes per token (vbm25_query_bytes)
        FL_TRY(d_bnd.alloc(4ull * (n_terms + 1)));
        gather_u32_kernel<<<(n_terms + 1 + 255) / 256, 256>>>(n_terms + 1, ds->d_term_first_block.as<uint32_t>(), a.off8, d_bnd.as<uint32_t>());
        FL_TRY(hipGetLastError());
        std::vector<uint32_t> bnd(size_t(n_terms) + 1);
        FL_TRY(hipMemcpy(bnd.data(), d_bnd.p, 4ull * (n_terms + 1), hipMemcpyDeviceToHost));
        for (uint32_t t = 0; t < n_terms; ++t)
            ds->term_bytes[t] = 8ull * (bnd[t + 1] - bnd[t]) + 40ull * (ds->term_first_block[t + 1] - ds->term_first_block[t]) + ds->term_df[t];
    } else {
        FL_TRY(hipMemset(ds->d_blk_off8.p, 0, 4));
    }
    FL_TRY(hipDeviceSynchronize());
    out = std::move(ds);
    return VBM25_OK;
}

// the host copy of a device segment: the same vbm25_segment the host builder makes, byte for byte
int download_device_segment(const vbm25_device_segment &ds, vbm25_segment **out) {
    FL_TRY(hipSetDevice(ds.device));
    auto seg = std::make_unique<vbm25_segment>();
    seg->k1 = ds.k1;
    seg->b = ds.b;
    seg->n_docs = ds.n_docs;
    seg->n_terms = ds.n_terms;
    seg->n_blocks = ds.n_blocks;
    seg->sum_len = ds.sum_len;
    seg->term_key = ds.term_key;
    seg->term_first_block = ds.term_first_block;
    seg->token_term = ds.token_term;
    auto fetch = [&](auto &vec, const HbmArray &src, size_t n) -> hipError_t {
        vec.resize(n);
        return n ? hipMemcpy(vec.data(), src.p, n * sizeof(vec[0]), hipMemcpyDeviceToHost) : hipSuccess;
    };
    FL_TRY(fetch(seg->doc_fieldnorm, ds.d_doc_fieldnorm, ds.n_docs));
    FL_TRY(fetch(seg->doc_payload, ds.d_doc_payload, 3ull * ds.n_docs));
    FL_TRY(fetch(seg->blk_off8, ds.d_blk_off8, size_t(ds.n_blocks) + 1));
    FL_TRY(fetch(seg->blob, ds.d_blob, ds.blob_bytes));
    FL_TRY(fetch(seg->blk_min_doc, ds.d_blk_min, ds.n_blocks));
    FL_TRY(fetch(seg->blk_max_doc, ds.d_blk_max, ds.n_blocks));
    FL_TRY(fetch(seg->blk_wand_tf, ds.d_blk_wand_tf, ds.n_blocks));
    FL_TRY(fetch(seg->blk_n, ds.d_blk_n, ds.n_blocks));
    FL_TRY(fetch(seg->blk_wand_fn, ds.d_blk_wand_fn, ds.n_blocks));
    FL_TRY(fetch(seg->blk_meta_doc, ds.d_blk_meta_doc, ds.n_blocks));
    FL_TRY(fetch(seg->blk_meta_tf, ds.d_blk_meta_tf, ds.n_blocks));
    FL_TRY(fetch(seg->term_wand_fn, ds.d_term_wand_fn, ds.n_terms));
    FL_TRY(fetch(seg->term_wand_tf, ds.d_term_wand_tf, ds.n_terms));
    FL_TRY(fetch(seg->term_df, ds.d_term_df, ds.n_terms));
    *out = seg.release();
    return VBM25_OK;
}

// dev_doc / dev_tf != nullptr: the mappings are already on the device, sorted (vbm25_segment_build_device_unsorted);
// post_doc / post_tf are not read then.
int build_device_impl(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len, const uint16_t *doc_payload,
                      uint32_t n_terms, const uint8_t *term_key, const uint64_t *term_start, const uint32_t *post_doc,
                      const uint32_t *post_tf, vbm25_segment **out, const uint32_t *dev_doc = nullptr, const uint32_t *dev_tf = nullptr) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!doc_len || !doc_payload) return set_error(VBM25_ERR_INVALID, "NULL argument");
    std::unique_ptr<vbm25_device_segment> ds;
    if (int rc = build_device_core(device, k1, b, n_docs, doc_len, nullptr, doc_payload, n_terms, term_key, term_start, post_doc, post_tf,
                                   dev_doc, dev_tf, ds))
        return rc;
    return download_device_segment(*ds, out);
}

// ---------------------------------------------------------------------------
// The synthetic corpus of SURVEY section 8(d) generated ON the device (the model of vbm25_segment_synth, segment.cpp: every
// draw slot of a document is token t with probability p_t, independently per token -- geometric gaps; tf = hits inside one
// document; document length = sum of its tfs) and sealed there: no posting crosses the PCIe link.  The same counter-based
// generator (splitmix64 seeded per (token, 65536-document chunk)) with the device's log / exp / cos: a corpus of the same
// distribution, NOT bit for bit the host generator's (libm and ocml round differently in a handful of draws per billion).
// ---------------------------------------------------------------------------
struct DevRng {
    unsigned long long s;
    __device__ explicit DevRng(unsigned long long seed) : s(seed) {}
    __device__ unsigned long long next() {  // splitmix64
        unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    __device__ double unit() { return ((double)(next() >> 11) + 1.0) * (1.0 / 9007199254740992.0); }  // (0,1]
};
__host__ __device__ inline unsigned long long synth_mix(unsigned long long a, unsigned long long b, unsigned long long c) {
    unsigned long long s = a ^ (b * 0xD6E8FEB86659FD93ull) ^ (c * 0xCA5A826395121157ull);
    unsigned long long z = 0;
    for (int i = 0; i < 2; ++i) {
        z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
    }
    return z;
}
constexpr uint32_t SYNTH_CHUNK = 1u << 16;

// draws per document: mean_len, or clamp(round(LogNormal(ln(0.8 mean_len), 0.6)), 8, 2000) (two uniforms per document)
__global__ void __launch_bounds__(256) synth_draws_kernel(uint32_t n_docs, uint32_t mean_len, uint32_t len_mode, unsigned long long seed0,
                                                          uint32_t *draws) {
    const double mu = log(0.8 * (double)mean_len);
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += gridDim.x * blockDim.x) {
        uint32_t len = mean_len;
        if (len_mode == 1) {
            DevRng rng(seed0 + 2ull * d * 0x9E3779B97F4A7C15ull);  // the state after 2 d draws of the sequential generator
            const double u1 = rng.unit(), u2 = rng.unit();
            const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
            const double v = nearbyint(exp(mu + 0.6 * z));
            len = (uint32_t)fmin(2000.0, fmax(8.0, v));
        }
        draws[d] = len;
    }
}

// A generation task is one token over one range of documents with its own random stream: a chunk of SYNTH_CHUNK documents, or
// -- for the head tokens of a Zipf law, whose chunk would be a serial loop of 10^5 draws in one lane while the rest of the
// device waits (C5: 14 s) -- one of 2^sub_log2 equal parts of it (about SYNTH_SUB_POSTINGS draws each).  Tokens in key order,
// a token's tasks in document order: the mappings come out sorted by (token, document).
constexpr uint32_t SYNTH_SUB_POSTINGS = 4096, SYNTH_SUB_MAX_LOG2 = 10;  // parts of >= 64 documents
struct SynthArgs {
    uint32_t n_docs, vocab, n_chunks;
    const unsigned long long *task_base;  // vocab + 1: first task of every key position
    const uint8_t *sub_log2;              // per key position
    unsigned long long n_tasks;
    unsigned long long seed;
    const unsigned long long *slot;  // n_docs + 1: prefix sum of the draws per document
    const double *log1mp;            // per token: log(1 - p_t)
    const uint32_t *order;           // key position -> token number
    uint32_t *count;                 // per (key position, chunk): postings
    const unsigned long long *offset;  // their exclusive prefix sum
    uint32_t *doc_len;               // sum of tf per document
    uint32_t *post_doc, *post_tf;
};
template <bool EMIT>
__global__ void __launch_bounds__(256) synth_gen_kernel(SynthArgs a) {
    const unsigned long long task = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (task >= a.n_tasks) return;
    uint32_t pos_i = 0;
    {   // the key position whose tasks hold this one: last i with task_base[i] <= task
        uint32_t lo_i = 0, hi_i = a.vocab;
        while (hi_i - lo_i > 1) {
            const uint32_t mid = (lo_i + hi_i) >> 1;
            if (a.task_base[mid] <= task) lo_i = mid; else hi_i = mid;
        }
        pos_i = lo_i;
    }
    const uint32_t sl2 = a.sub_log2[pos_i];
    const unsigned long long local = task - a.task_base[pos_i];
    const uint32_t chunk = (uint32_t)(local >> sl2), sub = (uint32_t)(local & ((1u << sl2) - 1u)), part = SYNTH_CHUNK >> sl2;
    const uint32_t token = a.order[pos_i];
    const uint32_t c0 = (uint32_t)min((unsigned long long)a.n_docs, (unsigned long long)chunk * SYNTH_CHUNK + (unsigned long long)sub * part);
    const uint32_t c1 = (uint32_t)min((unsigned long long)a.n_docs, (unsigned long long)c0 + part);
    if (c0 >= c1) {
        if (!EMIT) a.count[task] = 0;
        return;
    }
    const double l1p = a.log1mp[token];
    const unsigned long long *S = a.slot;
    DevRng rng(synth_mix(a.seed, token, c0));
    const unsigned long long begin = S[c0], end = S[c1];
    unsigned long long pos = begin, out = EMIT ? a.offset[task] : 0ull;
    uint32_t d = c0, cur_doc = 0xffffffffu, cur_tf = 0, n = 0;
    for (;;) {
        const double g = floor(log(rng.unit()) / l1p);
        if (!(g < 1e18)) break;
        pos += (unsigned long long)g;
        if (pos >= end) break;
        if (S[d + 1] <= pos) {  // slot -> document: interpolate inside the chunk, then walk
            const uint32_t guess = c0 + (uint32_t)((pos - begin) * (unsigned long long)(c1 - c0) / (end - begin));
            if (guess > d) d = guess;
            while (S[d] > pos) --d;
            while (S[d + 1] <= pos) ++d;
        }
        if (d == cur_doc) {
            ++cur_tf;
        } else {
            if (cur_tf) {
                if (EMIT) {
                    a.post_doc[out] = cur_doc;
                    a.post_tf[out] = cur_tf;
                    ++out;
                } else {
                    atomicAdd(&a.doc_len[cur_doc], cur_tf);
                    ++n;
                }
            }
            cur_doc = d;
            cur_tf = 1;
        }
        ++pos;
    }
    if (cur_tf) {
        if (EMIT) {
            a.post_doc[out] = cur_doc;
            a.post_tf[out] = cur_tf;
        } else {
            atomicAdd(&a.doc_len[cur_doc], cur_tf);
            ++n;
        }
    }
    if (!EMIT) a.count[task] = n;
}
__global__ void __launch_bounds__(256) gather_u64_at_kernel(uint32_t n, const unsigned long long *at, const unsigned long long *src,
                                                            unsigned long long last, unsigned long long *dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[at[i]];
    if (i == n) dst[i] = last;
}

int synth_device_impl(const vbm25_synth_params *pr, int device, vbm25_device_segment **out) {
    if (!pr || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (!pr->n_docs || !pr->vocab || !pr->mean_len) return set_error(VBM25_ERR_INVALID, "n_docs, vocab and mean_len must be positive");
    if (!(pr->k1 >= 1.2 && pr->k1 <= 2.0) || !(pr->b >= 0.0 && pr->b <= 1.0))
        return set_error(VBM25_ERR_INVALID, "k1 must be in [1.2, 2] and b in [0, 1]");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return set_error(VBM25_ERR_DEVICE, "no HIP device: the device generator has no CPU fallback (vbm25_segment_synth is the host generator)");
    if (device < 0 || device >= n_dev) return set_error(VBM25_ERR_INVALID, "device %d out of range (%d devices)", device, n_dev);
    FL_TRY(hipSetDevice(device));
    const uint32_t n_docs = pr->n_docs, vocab = pr->vocab, n_chunks = (n_docs + SYNTH_CHUNK - 1) / SYNTH_CHUNK;
    // token probabilities and the tokens in key order (bytewise order of their decimal strings): vocabulary-sized, on the host
    std::vector<double> log1mp(vocab);
    {
        double norm = 0.0;
        if (pr->zipf_s > 0)
            for (uint32_t t = 0; t < vocab; ++t) norm += std::pow(double(t + 1), -pr->zipf_s);
        for (uint32_t t = 0; t < vocab; ++t) {
            const double p = pr->zipf_s > 0 ? std::pow(double(t + 1), -pr->zipf_s) / norm : 1.0 / double(vocab);
            log1mp[t] = std::log1p(-std::min(p, 0.999999));
        }
    }
    std::vector<uint8_t> keys(16ull * vocab, 0);
    for (uint32_t t = 0; t < vocab; ++t) {
        char buf[17];
        const int n = std::snprintf(buf, sizeof buf, "%u", t);
        std::memcpy(keys.data() + 16ull * t, buf, size_t(n));
    }
    std::vector<uint32_t> order(vocab);
    for (uint32_t t = 0; t < vocab; ++t) order[t] = t;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return std::memcmp(keys.data() + 16ull * x, keys.data() + 16ull * y, 16) < 0; });
    // tasks: one per (key position, chunk), or 2^sub_log2 per chunk where a chunk of the token holds more than
    // 2 x SYNTH_SUB_POSTINGS draws on average (slots of a chunk ~ chunk documents x the mean draws per document)
    std::vector<unsigned long long> task_base(size_t(vocab) + 1);
    std::vector<uint8_t> sub_log2(vocab);
    const double slots_per_chunk = double(std::min<uint32_t>(SYNTH_CHUNK, n_docs)) * double(pr->mean_len);
    unsigned long long n_tasks = 0;
    for (uint32_t i = 0; i < vocab; ++i) {
        const double expect = -std::expm1(log1mp[order[i]]) * slots_per_chunk;  // p_t x slots
        uint32_t l2 = 0;
        while (l2 < SYNTH_SUB_MAX_LOG2 && expect > 2.0 * SYNTH_SUB_POSTINGS * double(1u << l2)) ++l2;
        sub_log2[i] = uint8_t(l2);
        task_base[i] = n_tasks;
        n_tasks += (unsigned long long)n_chunks << l2;
    }
    task_base[vocab] = n_tasks;
    if (n_tasks > 0x7fffffffull) return set_error(VBM25_ERR_UNSUPPORTED, "more than 2^31 generation tasks");

    DBuf d_draws, d_slot, d_l1p, d_order, d_count, d_off, d_len, d_pd, d_pt, d_tmp, d_bnd, d_tbase, d_sub;
    FL_TRY(d_draws.alloc(4ull * n_docs));
    FL_TRY(d_slot.alloc(8ull * (n_docs + 1ull)));
    FL_TRY(d_l1p.alloc(8ull * vocab));
    FL_TRY(d_order.alloc(4ull * vocab));
    FL_TRY(d_count.alloc(4ull * n_tasks));
    FL_TRY(d_off.alloc(8ull * n_tasks));
    FL_TRY(d_len.alloc(4ull * n_docs));
    FL_TRY(d_bnd.alloc(8ull * (vocab + 1ull)));
    FL_TRY(hipMemcpy(d_l1p.p, log1mp.data(), 8ull * vocab, hipMemcpyHostToDevice));
    FL_TRY(hipMemcpy(d_order.p, order.data(), 4ull * vocab, hipMemcpyHostToDevice));
    FL_TRY(d_tbase.alloc(8ull * (vocab + 1ull)));
    FL_TRY(d_sub.alloc(vocab));
    FL_TRY(hipMemcpy(d_tbase.p, task_base.data(), 8ull * (vocab + 1ull), hipMemcpyHostToDevice));
    FL_TRY(hipMemcpy(d_sub.p, sub_log2.data(), vocab, hipMemcpyHostToDevice));
    FL_TRY(hipMemset(d_len.p, 0, 4ull * n_docs));
    FL_TRY(hipMemset(d_slot.p, 0, 8));
    synth_draws_kernel<<<2048, 256>>>(n_docs, pr->mean_len, pr->len_mode, synth_mix(pr->seed, 0xD0C5, 0), d_draws.as<uint32_t>());
    FL_TRY(hipGetLastError());
    {   // slot[d + 1] = draws[0] + ... + draws[d]
        size_t tb = 0;
        hipcub::TransformInputIterator<unsigned long long, WidenU32, const uint32_t *> wide(d_draws.as<uint32_t>(), WidenU32());
        FL_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tb, wide, d_slot.as<unsigned long long>() + 1, (int)n_docs));
        FL_TRY(d_tmp.alloc(tb));
        FL_TRY(hipcub::DeviceScan::InclusiveSum(d_tmp.p, tb, wide, d_slot.as<unsigned long long>() + 1, (int)n_docs));
    }
    SynthArgs a{};
    a.n_docs = n_docs;
    a.vocab = vocab;
    a.n_chunks = n_chunks;
    a.task_base = d_tbase.as<unsigned long long>();
    a.sub_log2 = d_sub.as<uint8_t>();
    a.n_tasks = n_tasks;
    a.seed = pr->seed;
    a.slot = d_slot.as<unsigned long long>();
    a.log1mp = d_l1p.as<double>();
    a.order = d_order.as<uint32_t>();
    a.count = d_count.as<uint32_t>();
    a.offset = d_off.as<unsigned long long>();
    a.doc_len = d_len.as<uint32_t>();
    const uint32_t grid = (uint32_t)((n_tasks + 255) / 256);
    synth_gen_kernel<false><<<grid, 256>>>(a);  // pass 1: postings per task, document lengths
    FL_TRY(hipGetLastError());
    unsigned long long n_post = 0;
    {
        size_t tb = 0;
        hipcub::TransformInputIterator<unsigned long long, WidenU32, const uint32_t *> wide(d_count.as<uint32_t>(), WidenU32());
        DBuf d_t2;
        FL_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, wide, d_off.as<unsigned long long>(), (int)n_tasks));
        FL_TRY(d_t2.alloc(tb));
        FL_TRY(hipcub::DeviceScan::ExclusiveSum(d_t2.p, tb, wide, d_off.as<unsigned long long>(), (int)n_tasks));
        unsigned long long last_off = 0;
        uint32_t last_cnt = 0;
        FL_TRY(hipMemcpy(&last_off, d_off.as<unsigned long long>() + (n_tasks - 1), 8, hipMemcpyDeviceToHost));
        FL_TRY(hipMemcpy(&last_cnt, d_count.as<uint32_t>() + (n_tasks - 1), 4, hipMemcpyDeviceToHost));
        n_post = last_off + last_cnt;
    }
    // the first mapping of every token (key order); tokens that never occur are left out of the segment
    gather_u64_at_kernel<<<(vocab + 1 + 255) / 256, 256>>>(vocab, d_tbase.as<unsigned long long>(), d_off.as<unsigned long long>(), n_post, d_bnd.as<unsigned long long>());
    FL_TRY(hipGetLastError());
    std::vector<uint64_t> bnd(size_t(vocab) + 1);
    FL_TRY(hipMemcpy(bnd.data(), d_bnd.p, 8ull * (vocab + 1ull), hipMemcpyDeviceToHost));
    std::vector<uint64_t> term_start;
    std::vector<uint8_t> term_key;
    std::vector<uint32_t> token_term(vocab, UINT32_MAX);
    for (uint32_t i = 0; i < vocab; ++i) {
        if (bnd[i + 1] == bnd[i]) continue;
        token_term[order[i]] = uint32_t(term_start.size());
        term_start.push_back(bnd[i]);
        term_key.insert(term_key.end(), keys.begin() + 16ull * order[i], keys.begin() + 16ull * order[i] + 16);
    }
    const uint32_t n_terms = uint32_t(term_start.size());
    term_start.push_back(n_post);
    FL_TRY(d_pd.alloc(4ull * n_post));
    FL_TRY(d_pt.alloc(4ull * n_post));
    a.post_doc = d_pd.as<uint32_t>();
    a.post_tf = d_pt.as<uint32_t>();
    synth_gen_kernel<true><<<grid, 256>>>(a);  // pass 2: the mappings, in (token, document) order
    FL_TRY(hipGetLastError());
    // the generation buffers go before the encode's are made
    (void)hipFree(d_count.p); d_count.p = nullptr;
    (void)hipFree(d_off.p); d_off.p = nullptr;
    (void)hipFree(d_slot.p); d_slot.p = nullptr;
    (void)hipFree(d_draws.p); d_draws.p = nullptr;
    std::unique_ptr<vbm25_device_segment> ds;
    if (int rc = build_device_core(device, pr->k1, pr->b, n_docs, nullptr, d_len.as<uint32_t>(), nullptr, n_terms, term_key.data(), term_start.data(),
                                   nullptr, nullptr, d_pd.as<uint32_t>(), d_pt.as<uint32_t>(), ds))
        return rc;
    ds->token_term = std::move(token_term);
    *out = ds.release();
    return VBM25_OK;
}

// ---------------------------------------------------------------------------
// Mappings in any order (segment.rs:41-45: the sealed segment wants them by (token, document); the reference gets there
// with a k-way merge of sorted runs, io.rs:244-282): one radix sort of the 64-bit keys token << 32 | document on the
// device, the term frequencies riding along, then the same encode.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mapping_keys_kernel(uint64_t n, uint32_t n_terms, const uint32_t *term, const uint32_t *doc,
                                                           unsigned long long *key, uint32_t *error_flag) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t t = term[i];
        if (t >= n_terms) {
            atomicOr(error_flag, 1u);
            t = n_terms ? n_terms - 1u : 0u;  // (clamped: nothing downstream may index with it; the host rejects the input)
        }
        key[i] = (unsigned long long)t << 32 | doc[i];
    }
}
// sorted keys -> document column + the first mapping of every token (CSR starts)
__global__ void __launch_bounds__(256) mapping_split_kernel(uint64_t n, uint32_t n_terms, const unsigned long long *key, uint32_t *doc,
                                                            unsigned long long *term_start) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = key[i];
        doc[i] = (uint32_t)k;
        const uint32_t t = (uint32_t)(k >> 32);
        if (t < n_terms && (i == 0 || (uint32_t)(key[i - 1] >> 32) != t)) term_start[t] = i;
    }
}

int build_device_unsorted_impl(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len, const uint16_t *doc_payload,
                               uint32_t n_terms, const uint8_t *term_key, uint64_t n_map, const uint32_t *map_term,
                               const uint32_t *map_doc, const uint32_t *map_tf, vbm25_segment **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_map && (!map_term || !map_doc || !map_tf)) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (n_map > 0x7fffffffull) return set_error(VBM25_ERR_UNSUPPORTED, "more than 2^31 mappings in one sort");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return set_error(VBM25_ERR_DEVICE, "no HIP device: the device builder has no CPU fallback (vbm25_segment_build is the host builder)");
    if (device < 0 || device >= n_dev) return set_error(VBM25_ERR_INVALID, "device %d out of range (%d devices)", device, n_dev);
    FL_TRY(hipSetDevice(device));
    DBuf d_term, d_doc, d_tf, d_tf2, d_key, d_key2, d_tmp, d_ts, d_err;
    FL_TRY(d_term.alloc(4ull * n_map));
    FL_TRY(d_doc.alloc(4ull * n_map));
    FL_TRY(d_tf.alloc(4ull * n_map));
    FL_TRY(d_tf2.alloc(4ull * n_map));
    FL_TRY(d_key.alloc(8ull * n_map));
    FL_TRY(d_key2.alloc(8ull * n_map));
    FL_TRY(d_ts.alloc(8ull * (n_terms + 1ull)));
    FL_TRY(d_err.alloc(4));
    FL_TRY(hipMemset(d_err.p, 0, 4));
    FL_TRY(hipMemset(d_ts.p, 0xff, 8ull * (n_terms + 1ull)));
    std::vector<uint64_t> term_start(size_t(n_terms) + 1, n_map);
    if (n_map) {
        FL_TRY(hipMemcpy(d_term.p, map_term, 4ull * n_map, hipMemcpyHostToDevice));
        FL_TRY(hipMemcpy(d_doc.p, map_doc, 4ull * n_map, hipMemcpyHostToDevice));
        FL_TRY(hipMemcpy(d_tf.p, map_tf, 4ull * n_map, hipMemcpyHostToDevice));
        mapping_keys_kernel<<<2048, 256>>>(n_map, n_terms, d_term.as<uint32_t>(), d_doc.as<uint32_t>(), d_key.as<unsigned long long>(),
                                           d_err.as<uint32_t>());
        FL_TRY(hipGetLastError());
        {   // a token rank >= n_terms: rejected BEFORE the sort and the split (term_start has n_terms + 1 entries)
            uint32_t bad = 0;
            FL_TRY(hipMemcpy(&bad, d_err.p, 4, hipMemcpyDeviceToHost));
            if (bad) return set_error(VBM25_ERR_INVALID, "a mapping names a token >= n_terms");
        }
        int end_bit = 32;  // the bits of the key that can differ: the document and as much of the token as n_terms needs
        while (end_bit < 64 && (uint64_t(n_terms) >> (end_bit - 32)) != 0) ++end_bit;
        hipcub::DoubleBuffer<unsigned long long> keys(d_key.as<unsigned long long>(), d_key2.as<unsigned long long>());
        hipcub::DoubleBuffer<uint32_t> vals(d_tf.as<uint32_t>(), d_tf2.as<uint32_t>());
        size_t tmp_bytes = 0;
        FL_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, vals, (int)n_map, 0, end_bit));
        FL_TRY(d_tmp.alloc(tmp_bytes));
        FL_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tmp_bytes, keys, vals, (int)n_map, 0, end_bit));
        mapping_split_kernel<<<2048, 256>>>(n_map, n_terms, keys.Current(), d_doc.as<uint32_t>(), d_ts.as<unsigned long long>());
        FL_TRY(hipGetLastError());
        uint32_t flag = 0;
        FL_TRY(hipMemcpy(&flag, d_err.p, 4, hipMemcpyDeviceToHost));
        if (flag) return set_error(VBM25_ERR_INVALID, "a mapping names a token >= n_terms");
        FL_TRY(hipMemcpy(term_start.data(), d_ts.p, 8ull * n_terms, hipMemcpyDeviceToHost));
        term_start[n_terms] = n_map;
        for (uint32_t t = 0; t < n_terms; ++t)
            if (term_start[t] == ~0ull) return set_error(VBM25_ERR_INVALID, "term %u has no postings", t);
        // (a repeated (token, document) pair is found by the encode's validation: ids strictly ascending inside a token)
        return build_device_impl(device, k1, b, n_docs, doc_len, doc_payload, n_terms, term_key, term_start.data(), nullptr, nullptr, out,
                                 d_doc.as<uint32_t>(), vals.Current());
    }
    return build_device_impl(device, k1, b, n_docs, doc_len, doc_payload, n_terms, term_key, term_start.data(), map_doc, map_tf, out);
}

}  // namespace

extern "C" int vbm25_segment_build_device_unsorted(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                                                   const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                                                   uint64_t n_mappings, const uint32_t *map_term, const uint32_t *map_doc,
                                                   const uint32_t *map_tf, vbm25_segment **out) {
    return vbm25::guarded([&] {
        return build_device_unsorted_impl(device, k1, b, n_docs, doc_len, doc_payload, n_terms, term_key, n_mappings, map_term, map_doc,
                                          map_tf, out);
    });
}

extern "C" int vbm25_segment_build_device(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                                          const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                                          const uint64_t *term_start, const uint32_t *post_doc, const uint32_t *post_tf,
                                          vbm25_segment **out) {
    return vbm25::guarded([&] {
        return build_device_impl(device, k1, b, n_docs, doc_len, doc_payload, n_terms, term_key, term_start, post_doc, post_tf, out);
    });
}

extern "C" int vbm25_device_segment_synth(const vbm25_synth_params *params, int device, vbm25_device_segment **out) {
    return vbm25::guarded([&] { return synth_device_impl(params, device, out); });
}
extern "C" int vbm25_device_segment_build(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                                          const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key, const uint64_t *term_start,
                                          const uint32_t *post_doc, const uint32_t *post_tf, vbm25_device_segment **out) {
    return vbm25::guarded([&]() -> int {
        if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
        *out = nullptr;
        if (!doc_len || !doc_payload) return set_error(VBM25_ERR_INVALID, "NULL argument");
        std::unique_ptr<vbm25_device_segment> ds;
        if (int rc = build_device_core(device, k1, b, n_docs, doc_len, nullptr, doc_payload, n_terms, term_key, term_start, post_doc, post_tf,
                                       nullptr, nullptr, ds))
            return rc;
        *out = ds.release();
        return VBM25_OK;
    });
}
extern "C" int vbm25_device_segment_download(const vbm25_device_segment *ds, vbm25_segment **out) {
    return vbm25::guarded([&]() -> int {
        if (!ds || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
        *out = nullptr;
        return download_device_segment(*ds, out);
    });
}
extern "C" int vbm25_device_segment_token_terms(const vbm25_device_segment *ds, const uint32_t *tokens, uint32_t n, uint32_t *term_ids) {
    if (!ds || ds->token_term.empty() || (n && (!tokens || !term_ids)))
        return set_error(VBM25_ERR_INVALID, "not a segment of vbm25_device_segment_synth / NULL argument");
    for (uint32_t i = 0; i < n; ++i) term_ids[i] = tokens[i] < ds->token_term.size() ? ds->token_term[tokens[i]] : UINT32_MAX;
    return VBM25_OK;
}
extern "C" uint64_t vbm25_device_segment_query_bytes(const vbm25_device_segment *ds, const uint32_t *term_ids, uint32_t n_terms, uint32_t k) {
    uint64_t bytes = 0;
    if (ds)
        for (uint32_t i = 0; i < n_terms; ++i)
            if (term_ids[i] < ds->n_terms) bytes += ds->term_bytes[term_ids[i]];
    return bytes + 14ull * k;
}
extern "C" int vbm25_device_segment_info(const vbm25_device_segment *ds, uint32_t *n_docs, uint32_t *n_terms, uint32_t *n_blocks, uint64_t *n_postings) {
    if (!ds) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (n_docs) *n_docs = ds->n_docs;
    if (n_terms) *n_terms = ds->n_terms;
    if (n_blocks) *n_blocks = ds->n_blocks;
    if (n_postings) {
        *n_postings = 0;
        for (uint32_t df : ds->term_df) *n_postings += df;
    }
    return VBM25_OK;
}
extern "C" void vbm25_device_segment_free(vbm25_device_segment *ds) {
    if (!ds) return;
    (void)hipSetDevice(ds->device);
    delete ds;
}
