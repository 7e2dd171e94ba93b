// search.hip -- device half of libvbm25: batched BM25 top-k over compressed posting blocks
// for MI355X (gfx950, wave64).  Replaces the traversal of
// /root/reference/crates/bm25/src/search.rs:28-282 (bm25::search) for the sealed segment.
//
// Shape of the computation (DESIGN.md has the full story):
//   plan_kernel       one workgroup: splits every query into doc-range chunks of roughly equal
//                     posting counts -> work items (query, doc_lo, doc_hi, #terms | dense flag)
//   scan_kernel       sparse queries (<= 12 indexed terms): one workgroup per item walks the
//                     chunk in doc-range tiles with one barrier per tile.  Worker waves decode
//                     128-posting blocks (bit-unpack + DPP prefix sum, search.rs:498-518 /
//                     compression.rs:65-136), mark every posting in hashed LDS bitmaps and drop
//                     or score (Cache::evaluate, bm25.rs:355-358) documents with a single
//                     posting; a joiner wave adds up, in ascending key order (evaluate.rs:43-72),
//                     the documents whose postings collide; a planner wave plans the tiles from
//                     block metadata staged in LDS and owns the running top-k (Results,
//                     search.rs:284-314).  A per-query threshold is shared between workgroups
//                     through a 64-bit atomic max on the score bits.
//   scan_many_kernel  queries with many terms or many postings per document, and items the
//                     sparse kernel gave up on: term-phased accumulation in dense doc windows
//                     or an LDS hash table.
//   merge_kernel      one wave per query: merges the per-chunk top-k lists, adds payloads.
//
// Result order is canonical: score descending, ties by ascending doc id.  All f64 arithmetic
// is IEEE (compiled with -ffp-contract=off, no fast-math): results are bit-identical to the
// CPU oracle's brute-force evaluation.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "vbm25_internal.h"

namespace vbm25 {

static thread_local char g_error[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return set_error(VBM25_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,               \
                             hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

// ---------------------------------------------------------------------------
// Device-side view of the index and of one batch
// ---------------------------------------------------------------------------
struct DevIndex {
    uint32_t n_docs, n_terms, n_blocks;
    const uint32_t *term_df;
    const uint32_t *term_first_block;
    const double *term_s0;       // idf * (k1 + 1), host-computed (libm log)
    const uint32_t *term_wand_tf;  // TokenTuple WAND pair: the posting that maximises tf()
    const uint8_t *term_wand_fn;
    const uint32_t *blk_min_doc;
    const uint32_t *blk_max_doc;
    const uint4 *blk_meta;       // {min_doc, max_doc, off8, n | meta_doc<<8 | meta_tf<<16 | wand_fn<<24}
    const double *blk_ub;        // Cache::evaluate(block WAND pair) x (1 + 1e-12): no posting of the block scores higher
    const uint8_t *blob;
    const uint8_t *post_fn;      // derived: fieldnorm of every posting, 128 bytes per block
    const uint16_t *doc_payload;
    const double *s1;            // 256 entries
};

struct Item {
    uint32_t q, doc_lo, doc_hi, m;  // m = number of indexed terms of query q | ITEM_DENSE
};
constexpr uint32_t ITEM_DENSE = 0x80000000u;  // postings per document high: dense-window path

struct DevBatch {
    const uint32_t *term_ids;
    const uint32_t *q_off;
    uint32_t nq, k;
    Item *items;
    uint32_t *n_items;
    uint32_t *q_item_base;  // nq + 1
    unsigned long long *theta;  // per query: bits of a lower bound of the k-th best score
    double *res_score;      // per item: k entries
    uint32_t *res_doc;
    uint32_t *res_cnt;
    vbm25_hit *hits;
    uint32_t *n_hits;
    uint32_t *error_flag;
    const uint8_t *q_dense;    // per query: 1 = dense (many postings per document), host decided
    unsigned long long *spill; // per scan workgroup: candidates that did not fit the LDS buffer
    uint32_t *item_failed;     // per item: 1 = the chain kernel gave up (dense tile), redo it
    unsigned long long *prof;  // VBM25_PROFILE builds: 33 counters per workgroup
    uint32_t *work_ctr;        // next item of the cursor kernel (reset by plan_kernel)
    uint32_t *hist;            // per query: CUR_HB score buckets, documents accepted by any item
    uint32_t chain_min_terms;  // scan_kernel leaves queries with fewer terms to scan_cursor_kernel
};

constexpr int WG = 256;
constexpr int NW = WG / 64;
constexpr int SLOTS_LOG2 = 12;
constexpr int SLOTS = 1 << SLOTS_LOG2;  // hash table slots per workgroup
constexpr int CAP_BLOCKS = SLOTS / 2 / 128;  // blocks admitted per tile in hash mode
constexpr int MAX_TERMS = 128;         // terms per query handled on the GPU
constexpr uint32_t EMPTY = 0xffffffffu;
constexpr uint32_t TARGET_ITEMS = 1536;  // 2 x (256 CUs x 3 resident workgroups): measured best of 768..3072
constexpr uint32_t MIN_CHUNK_POSTINGS = 8192;
constexpr int PLAN_WG = 1024;
// chain kernel (scan_kernel) geometry
constexpr int CNW = 6;                   // worker waves per workgroup
constexpr int CWG = (CNW + 2) * 64;      // + one planner / merger wave + one joiner wave
constexpr int C_BLOCKS = 2 * CNW;        // block slots of staging per workgroup (2 per worker)
constexpr int C_POSTINGS = C_BLOCKS * 128;
constexpr int CHAIN_MAX_TERMS = C_BLOCKS;  // queries with more indexed terms use scan_many_kernel
constexpr int SLOW_CAP = 64;              // colliding postings per tile kept in LDS (rest: global spill)
constexpr int SLOW_ABORT = 512;           // beyond this the tile is dense: give the item to scan_many_kernel
constexpr int JC_CAP = 64;                // joined documents per tile kept in LDS (rest: global spill)
constexpr int CAND_CAP = 96;              // fast-path documents per tile kept in LDS (rest: global spill)
constexpr int BM_BITS_LOG2 = 14;          // hashed document bitmaps: 16384 bits each
constexpr int BM_WORDS = (1 << BM_BITS_LOG2) / 32;
constexpr int REG_K = 256;                // largest k whose running top-k lives in registers
constexpr uint32_t NONE32 = 0xffffffffu;

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
    uint32_t x = v0 + v1;  // inclusive scan of the per-lane sums
    const uint32_t own = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if ((int)lane >= o) x += y;
    }
    d0 = min_doc + (x - own) + v0;
    d1 = d0 + v1;
}

// ---------------------------------------------------------------------------
// Index preparation: fieldnorm of every posting + structural validation
// ---------------------------------------------------------------------------
struct PostFnArgs {
    uint32_t n_blocks, n_docs, n_terms;
    const uint4 *blk_meta;
    const uint8_t *blob, *doc_fieldnorm;
    uint8_t *post_fn;
    uint32_t *error_flag;
    // upper bounds to verify: the scan kernels prune with them
    const uint32_t *term_first_block, *term_wand_tf;
    const uint8_t *term_wand_fn;
    const double *term_s0, *s1, *blk_ub;
};
__global__ void __launch_bounds__(256) post_fn_kernel(PostFnArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t j = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (j >= a.n_blocks) return;
    const uint4 m = a.blk_meta[j];
    const uint32_t n = m.w & 0xff, md = (m.w >> 8) & 0xff, mt = (m.w >> 16) & 0xff;
    const uint8_t *body = a.blob + 8ull * m.z;
    uint32_t d0, d1;
    decode_doc_ids(body, md, n, m.x, lane, d0, d1);
    const uint32_t i0 = 2 * lane, i1 = i0 + 1;
    uint8_t f0 = 0, f1 = 0;
    bool bad = false;
    if (i0 < n) {
        bad |= d0 >= a.n_docs;
        if (d0 < a.n_docs) f0 = a.doc_fieldnorm[d0];
    }
    if (i1 < n) {
        bad |= d1 >= a.n_docs || d1 <= d0;
        if (d1 < a.n_docs) f1 = a.doc_fieldnorm[d1];
    }
    // strictly increasing across lanes, first = min_doc, last = max_doc
    const uint32_t prev = __shfl_up(d1, 1);
    if (lane > 0 && i0 < n) bad |= d0 <= prev;
    if (i0 == 0) bad |= d0 != m.x;
    if (i0 == n - 1) bad |= d0 != m.y;
    if (i1 == n - 1) bad |= d1 != m.y;
    if (bad) atomicOr(a.error_flag, 1u);
    reinterpret_cast<uchar2 *>(a.post_fn + 128ull * j)[lane] = make_uchar2(f0, f1);

    // The WAND pairs must bound every posting: Cache::evaluate of each posting against the block's
    // bound (blk_ub, margin included) and the token's (search.rs:363,377-380).
    uint32_t lo = 0, hi = a.n_terms;  // the term of block j: term_first_block[t] <= j < [t + 1]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.term_first_block[mid] <= j) lo = mid; else hi = mid;
    }
    const double s0 = a.term_s0[lo];
    const double wtf = (double)a.term_wand_tf[lo];
    const double tub = ((wtf * s0) / (wtf + a.s1[a.term_wand_fn[lo]])) * (1.0 + 1e-12);
    const double bub = a.blk_ub[j];
    uint32_t t0, t1;
    decode_fields(body + ((payload_bytes(md, n) + 7u) & ~7u), mt, n, lane, t0, t1);
    bool loose = false;
    if (i0 < n) {
        const double tf = (double)t0, p = (tf * s0) / (tf + a.s1[f0]);
        loose |= p > bub || p > tub;
    }
    if (i1 < n) {
        const double tf = (double)t1, p = (tf * s0) / (tf + a.s1[f1]);
        loose |= p > bub || p > tub;
    }
    if (loose) atomicOr(a.error_flag, 2u);
}

// ---------------------------------------------------------------------------
// Planner
// ---------------------------------------------------------------------------
// Block-wide inclusive scan of one u64 per thread (PLAN_WG threads): wave scans + one LDS hop.
__device__ __forceinline__ unsigned long long plan_incl_scan(unsigned long long v, unsigned long long *s_wave,
                                                             unsigned long long &total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(v, o);
        if ((int)lane >= o) v += y;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (uint32_t w = 0; w < PLAN_WG / 64; ++w) {
        const unsigned long long x = s_wave[w];
        if (w < wave) before += x;
        all += x;
    }
    __syncthreads();
    total = all;
    return v + before;
}

__global__ void __launch_bounds__(PLAN_WG) plan_kernel(DevIndex ix, DevBatch bt, uint32_t max_items,
                                                       uint32_t target_items, uint32_t min_chunk) {
    __shared__ unsigned long long s_wave[PLAN_WG / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (bt.nq + PLAN_WG - 1) / PLAN_WG;
    const uint32_t q0 = min(bt.nq, tid * per), q1 = min(bt.nq, q0 + per);
    // per-launch state of the scan kernels (saves two memset launches per step)
    for (uint32_t i = tid; i < bt.nq; i += PLAN_WG) bt.theta[i] = 0;
    for (uint32_t i = tid; i < max_items; i += PLAN_WG) bt.item_failed[i] = 0;
    if (tid == 0) *bt.work_ctr = 0;

    auto postings_of = [&](uint32_t q) {
        unsigned long long t = 0;
        for (uint32_t p = bt.q_off[q]; p < bt.q_off[q + 1]; ++p) {
            uint32_t term = bt.term_ids[p];
            if (term < ix.n_terms) t += ix.term_df[term];
        }
        return t;
    };
    unsigned long long local = 0;
    for (uint32_t q = q0; q < q1; ++q) local += postings_of(q);
    unsigned long long total = 0;
    plan_incl_scan(local, s_wave, total);
    unsigned long long chunk = (total + target_items - 1) / target_items;
    if (chunk < min_chunk) chunk = min_chunk;
    auto chunks_of = [&](uint32_t q) -> uint32_t {
        unsigned long long t = postings_of(q);
        if (t == 0) return 0u;
        // nearest, not ceil: a batch of similar queries gets the same count for all of them, i.e. the
        // item count lands on the target (a multiple of the resident waves) instead of ~8 % above it
        unsigned long long c = (t + chunk / 2) / chunk;
        if (c == 0) c = 1;
        if (c > ix.n_docs) c = ix.n_docs;
        return (uint32_t)c;
    };
    unsigned long long cnt = 0;
    for (uint32_t q = q0; q < q1; ++q) cnt += chunks_of(q);
    unsigned long long run = 0;
    const unsigned long long incl = plan_incl_scan(cnt, s_wave, run);
    if (tid == 0) {
        *bt.n_items = (uint32_t)min(run, (unsigned long long)max_items);
        if (run > max_items) atomicOr(bt.error_flag, 2u);
        bt.q_item_base[bt.nq] = (uint32_t)min(run, (unsigned long long)max_items);
    }
    uint32_t base = (uint32_t)(incl - cnt);
    for (uint32_t q = q0; q < q1; ++q) {
        const uint32_t c = chunks_of(q);
        uint32_t nterms = 0;
        for (uint32_t p = bt.q_off[q]; p < bt.q_off[q + 1]; ++p) nterms += bt.term_ids[p] < ix.n_terms;
        if (bt.q_dense[q]) nterms |= ITEM_DENSE;
        bt.q_item_base[q] = min(base, max_items);
        for (uint32_t i = 0; i < c && base + i < max_items; ++i) {
            Item it;
            it.q = q;
            it.doc_lo = (uint32_t)((unsigned long long)ix.n_docs * i / c);
            it.doc_hi = (uint32_t)((unsigned long long)ix.n_docs * (i + 1) / c);
            it.m = nterms;
            bt.items[base + i] = it;
        }
        base += c;
    }
}

// ---------------------------------------------------------------------------
// Sorted top-k list in LDS, maintained by ONE wave.
// Order: score descending, then doc id ascending ("better").
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool better(double sa, uint32_t da, double sb, uint32_t db) {
    return sa > sb || (sa == sb && da < db);
}

template <int KMAX>
struct TopK {
    double score[KMAX];
    uint32_t doc[KMAX];
    uint32_t count;
};

// Wave-cooperative insert of (s, d); caller guarantees it qualifies.  All 64 lanes call.
template <int KMAX>
__device__ __forceinline__ void topk_insert(TopK<KMAX> &L, uint32_t k, double s, uint32_t d,
                                            uint32_t lane) {
    const uint32_t n = L.count;
    uint32_t c = 0;
    for (uint32_t i = lane; i < n; i += 64) c += better(L.score[i], L.doc[i], s, d) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    const uint32_t pos = c;
    const uint32_t newn = n < k ? n + 1 : k;
    if (pos >= newn) return;
    // shift [pos, newn-2] up by one, from the top down
    for (int base = (int)newn - 2; base >= (int)pos; base -= 64) {
        const int i = base - (int)lane;
        double ts = 0;
        uint32_t td = 0;
        const bool act = i >= (int)pos;
        if (act) {
            ts = L.score[i];
            td = L.doc[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (act) {
            L.score[i + 1] = ts;
            L.doc[i + 1] = td;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        L.score[pos] = s;
        L.doc[pos] = d;
        L.count = newn;
    }
    __builtin_amdgcn_wave_barrier();
}

// Offer up to 64 candidates (one per lane, `has` marks validity) to the list.
template <int KMAX>
__device__ __forceinline__ void topk_offer(TopK<KMAX> &L, uint32_t k, bool has, double s,
                                           uint32_t d, uint32_t lane) {
    for (;;) {
        const uint32_t n = L.count;
        bool alive = has;
        if (alive && n >= k) alive = better(s, d, L.score[k - 1], L.doc[k - 1]);
        const unsigned long long mask = __ballot(alive);
        if (!mask) break;
        const int leader = __ffsll((long long)mask) - 1;
        const double cs = __shfl(s, leader);
        const uint32_t cd = __shfl(d, leader);
        topk_insert<KMAX>(L, k, cs, cd, lane);
        if ((int)lane == leader) has = false;
    }
}

// ---------------------------------------------------------------------------
// Posting scan
// ---------------------------------------------------------------------------
template <int KMAX>
__global__ void __launch_bounds__(WG) scan_many_kernel(DevIndex ix, DevBatch bt) {
    __shared__ uint32_t s_key[SLOTS];
    __shared__ double s_val[SLOTS];
    __shared__ uint16_t s_cand[SLOTS];
    __shared__ double s_s1[256];
    __shared__ TopK<KMAX> s_top;
    __shared__ uint32_t t_cur[MAX_TERMS], t_end[MAX_TERMS], t_quota[MAX_TERMS];
    __shared__ double t_s0[MAX_TERMS];
    __shared__ uint32_t s_m, s_hi, s_next_lo, s_cand_cnt, s_dense;
    __shared__ unsigned long long s_theta, s_sumdf;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t k = bt.k;
    for (int i = tid; i < 256; i += WG) s_s1[i] = ix.s1[i];

    const uint32_t n_items = *bt.n_items;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const Item it = bt.items[item];
        const bool failed = it.m <= (uint32_t)CHAIN_MAX_TERMS && bt.item_failed[item] != 0;
        if (it.m <= (uint32_t)CHAIN_MAX_TERMS && !failed) continue;  // done by scan_kernel
        const bool force_dense = failed || (it.m & ITEM_DENSE) != 0;
        const uint32_t q = it.q, clo = it.doc_lo, chi = it.doc_hi;
        __syncthreads();  // previous item fully done with LDS
        if (tid == 0) {
            // valid terms of the query, ascending (Query::new guarantees sorted keys)
            uint32_t m = 0;
            unsigned long long sum = 0;
            for (uint32_t p = bt.q_off[q]; p < bt.q_off[q + 1]; ++p) {
                const uint32_t term = bt.term_ids[p];
                if (term >= ix.n_terms) continue;  // search.rs:59-61
                if (m < MAX_TERMS) {
                    t_cur[m] = term;  // resolved below
                    sum += ix.term_df[term];
                    ++m;
                }
            }
            s_m = m;
            s_sumdf = sum;
            s_top.count = 0;
            s_dense = (force_dense || m >= (uint32_t)CAP_BLOCKS) ? 1u : 0u;
        }
        __syncthreads();
        const uint32_t m = s_m;
        const bool dense = s_dense != 0;
        if (tid < m) {
            const uint32_t term = t_cur[tid];
            const uint32_t b0 = ix.term_first_block[term], b1 = ix.term_first_block[term + 1];
            // first block whose max_doc >= clo
            uint32_t lo = b0, hi = b1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ix.blk_max_doc[mid] < clo) lo = mid + 1; else hi = mid;
            }
            t_end[tid] = b1;
            t_s0[tid] = ix.term_s0[term];
            const unsigned long long df = ix.term_df[term];
            const uint32_t share = (uint32_t)(((unsigned long long)(CAP_BLOCKS - (dense ? 0 : (int)m)) * df) / s_sumdf);
            t_quota[tid] = share > 1 ? share : 1;
            t_cur[tid] = lo;
        }
        __syncthreads();

        uint32_t lo = clo;
        unsigned long long published = 0;
        while (lo < chi) {
            // ---- tile bounds + table reset
            if (tid == 0) {
                s_hi = dense ? (chi - lo > (uint32_t)SLOTS ? lo + SLOTS : chi) : chi;
                s_cand_cnt = 0;
                s_next_lo = chi;
                s_theta = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int i = tid; i < SLOTS; i += WG) s_key[i] = EMPTY;
            __syncthreads();
            if (!dense && tid < m) {
                const uint32_t j = t_cur[tid] + t_quota[tid];
                if (j < t_end[tid]) atomicMin(&s_hi, ix.blk_min_doc[j]);
            }
            __syncthreads();
            const uint32_t hi = s_hi;

            // ---- accumulate, one term per phase (ascending key order)
            for (uint32_t t = 0; t < m; ++t) {
                const uint32_t jend = t_end[t];
                const double s0 = t_s0[t];
                for (uint32_t j = t_cur[t] + wave; j < jend; j += NW) {
                    const uint4 bm = ix.blk_meta[j];
                    if (bm.x >= hi) break;
                    const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
                    const uint8_t *body = ix.blob + 8ull * bm.z;
                    uint32_t d0, d1, f0, f1;
                    decode_doc_ids(body, md, n, bm.x, lane, d0, d1);
                    decode_fields(body + ((payload_bytes(md, n) + 7u) & ~7u), mt, n, lane, f0, f1);
                    const uchar2 fn = reinterpret_cast<const uchar2 *>(ix.post_fn + 128ull * j)[lane];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const uint32_t i = 2 * lane + e;
                        const uint32_t d = e ? d1 : d0;
                        const uint32_t tfv = e ? f1 : f0;
                        const uint32_t f = e ? fn.y : fn.x;
                        if (i < n && d >= lo && d < hi) {
                            const double tf = (double)tfv;
                            const double p = (tf * s0) / (tf + s_s1[f]);  // bm25.rs:355-358
                            const uint32_t key = d - lo;
                            uint32_t slot = dense ? key : ((key * 0x9E3779B1u) >> (32 - SLOTS_LOG2));
                            for (;;) {
                                const uint32_t prev = atomicCAS(&s_key[slot], EMPTY, key);
                                if (prev == EMPTY) {
                                    s_val[slot] = p;
                                    break;
                                }
                                if (prev == key) {
                                    s_val[slot] += p;
                                    break;
                                }
                                slot = (slot + 1) & (SLOTS - 1);
                            }
                        }
                    }
                }
                __syncthreads();
            }

            // ---- candidates of this tile
            {
                const unsigned long long theta = s_theta;
                const uint32_t n = s_top.count;
                const double ws = n >= k ? s_top.score[k - 1] : 0.0;
                const uint32_t wd = n >= k ? s_top.doc[k - 1] : 0u;
                for (int i = tid; i < SLOTS; i += WG) {
                    const uint32_t key = s_key[i];
                    if (key == EMPTY) continue;
                    const double sc = s_val[i];
                    if ((unsigned long long)__double_as_longlong(sc) < theta) continue;
                    if (n >= k && !better(sc, lo + key, ws, wd)) continue;
                    const uint32_t at = atomicAdd(&s_cand_cnt, 1u);
                    s_cand[at] = (uint16_t)i;
                }
            }
            __syncthreads();
            if (wave == 0) {
                const uint32_t cnt = s_cand_cnt;
                for (uint32_t base = 0; base < cnt; base += 64) {
                    const bool has = base + lane < cnt;
                    double sc = 0;
                    uint32_t d = 0;
                    if (has) {
                        const uint32_t slot = s_cand[base + lane];
                        sc = s_val[slot];
                        d = lo + s_key[slot];
                    }
                    topk_offer<KMAX>(s_top, k, has, sc, d, lane);
                }
                if (s_top.count >= k && lane == 0) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(s_top.score[k - 1]);
                    if (bits > published) {
                        atomicMax(&bt.theta[q], bits);
                        published = bits;
                    }
                }
            }
            // ---- advance cursors; next tile starts at the first remaining posting
            if (tid < m) {
                uint32_t j = t_cur[tid];
                const uint32_t e = t_end[tid];
                while (j < e && ix.blk_max_doc[j] < hi) ++j;
                t_cur[tid] = j;
                if (j < e) atomicMin(&s_next_lo, max(hi, ix.blk_min_doc[j]));
            }
            __syncthreads();
            lo = max(hi, s_next_lo);
        }

        // ---- chunk result
        __syncthreads();
        {
            const uint32_t n = s_top.count;
            for (uint32_t i = tid; i < n; i += WG) {
                bt.res_score[(size_t)item * k + i] = s_top.score[i];
                bt.res_doc[(size_t)item * k + i] = s_top.doc[i];
            }
            if (tid == 0) bt.res_cnt[item] = n;
        }
    }
}


// ---------------------------------------------------------------------------
// Split decode used by scan_kernel: the two dwords that hold a field are fetched early
// (possibly one tile ahead) and the field is extracted later.  One formula covers every
// codec of compression.rs:65-136: bit-packed (lane stream words 16 bytes apart), width 32 /
// bytewidth 4 (raw), byte-packed tails (unaligned little-endian bytes).
// ---------------------------------------------------------------------------
struct FieldAddr {
    uint32_t off0, off1, sh, mask;
};
__device__ __forceinline__ FieldAddr field_addr(uint32_t meta, uint32_t n, uint32_t i) {
    FieldAddr a;
    const uint32_t width = meta & 127u;
    if ((meta >> 7) == 0) {
        if (width == 32) {
            a.off0 = a.off1 = 4 * i;
            a.sh = 0;
            a.mask = 0xffffffffu;
        } else {
            const uint32_t l = i & 3, bit = (i >> 2) * width, w = bit >> 5;
            a.sh = bit & 31;
            a.off0 = 16 * w + 4 * l;
            a.off1 = a.off0 + ((a.sh + width > 32) ? 16u : 0u);
            a.mask = (1u << width) - 1u;  // width 0 -> mask 0 -> field 0
        }
    } else {
        const uint32_t bo = (i < n ? i : 0u) * width;
        a.off0 = bo & ~3u;
        a.off1 = a.off0 + 4;
        a.sh = 8 * (bo & 3u);
        a.mask = width >= 4 ? 0xffffffffu : (1u << (8 * width)) - 1u;
    }
    return a;
}
__device__ __forceinline__ uint32_t field_val(uint32_t lo, uint32_t hi, const FieldAddr &a) {
    return __builtin_amdgcn_alignbit(hi, lo, a.sh) & a.mask;  // ((hi:lo) >> sh), sh < 32
}
struct BlockFetch {  // raw dwords of one block for this lane: doc fields 0/1, tf fields 0/1
    uint32_t dlo0, dhi0, dlo1, dhi1, tlo0, thi0, tlo1, thi1;
    uint32_t fn;  // two fieldnorm bytes
};
// Bit-packed blocks: a lane's two values (indices 2L, 2L+1) sit in adjacent lane streams at the
// same step, so their words are one aligned 8-byte pair in group w and one in group w+1.
__device__ __forceinline__ void pair_fetch(const uint8_t *__restrict__ p, uint32_t width, uint32_t lane,
                                           uint32_t &lo0, uint32_t &hi0, uint32_t &lo1, uint32_t &hi1) {
    const uint32_t bit = __umul24(lane >> 1, width);   // step t = (2L) >> 2
    const uint32_t off = 16 * (bit >> 5) + 8 * (lane & 1);  // streams l0 = 2*(L&1), l0 + 1
    const uint2 a = *reinterpret_cast<const uint2 *>(p + off);
    const uint2 b = *reinterpret_cast<const uint2 *>(p + off + 16);  // may be the next payload: unused then
    lo0 = a.x;
    lo1 = a.y;
    hi0 = b.x;
    hi1 = b.y;
}
__device__ __forceinline__ void pair_extract(uint32_t width, uint32_t lane, uint32_t lo0, uint32_t hi0,
                                             uint32_t lo1, uint32_t hi1, uint32_t &v0, uint32_t &v1) {
    const uint32_t sh = __umul24(lane >> 1, width) & 31;
    const uint32_t mask = width >= 32 ? 0xffffffffu : (1u << width) - 1u;
    v0 = __builtin_amdgcn_alignbit(hi0, lo0, sh) & mask;  // ((hi:lo) >> sh), sh < 32
    v1 = __builtin_amdgcn_alignbit(hi1, lo1, sh) & mask;
}
__device__ __forceinline__ void block_fetch(const DevIndex &ix, const uint4 bm, uint32_t j,
                                            uint32_t lane, BlockFetch &f) {
    const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
    const uint8_t *body = ix.blob + 8ull * bm.z;
    const uint8_t *tbody = body + ((payload_bytes(md, n) + 7u) & ~7u);
    if ((md >> 7) == 0) {  // full block (both streams bit-packed, compression.rs:42-52,99-103)
        pair_fetch(body, md & 127u, lane, f.dlo0, f.dhi0, f.dlo1, f.dhi1);
        pair_fetch(tbody, mt & 127u, lane, f.tlo0, f.thi0, f.tlo1, f.thi1);
    } else {               // tail block: byte-packed, generic addressing
        const FieldAddr a0 = field_addr(md, n, 2 * lane), a1 = field_addr(md, n, 2 * lane + 1);
        const FieldAddr b0 = field_addr(mt, n, 2 * lane), b1 = field_addr(mt, n, 2 * lane + 1);
        f.dlo0 = *reinterpret_cast<const uint32_t *>(body + a0.off0);
        f.dhi0 = *reinterpret_cast<const uint32_t *>(body + a0.off1);
        f.dlo1 = *reinterpret_cast<const uint32_t *>(body + a1.off0);
        f.dhi1 = *reinterpret_cast<const uint32_t *>(body + a1.off1);
        f.tlo0 = *reinterpret_cast<const uint32_t *>(tbody + b0.off0);
        f.thi0 = *reinterpret_cast<const uint32_t *>(tbody + b0.off1);
        f.tlo1 = *reinterpret_cast<const uint32_t *>(tbody + b1.off0);
        f.thi1 = *reinterpret_cast<const uint32_t *>(tbody + b1.off1);
    }
    f.fn = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * j)[lane];
}
// fields of a fetched block: document-id deltas (or raw ids) and term frequencies
__device__ __forceinline__ void block_fields(const uint4 bm, uint32_t lane, const BlockFetch &f,
                                             uint32_t &v0, uint32_t &v1, uint32_t &f0, uint32_t &f1) {
    const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
    if ((md >> 7) == 0) {
        pair_extract(md & 127u, lane, f.dlo0, f.dhi0, f.dlo1, f.dhi1, v0, v1);
        pair_extract(mt & 127u, lane, f.tlo0, f.thi0, f.tlo1, f.thi1, f0, f1);
    } else {
        v0 = field_val(f.dlo0, f.dhi0, field_addr(md, n, 2 * lane));
        v1 = field_val(f.dlo1, f.dhi1, field_addr(md, n, 2 * lane + 1));
        f0 = field_val(f.tlo0, f.thi0, field_addr(mt, n, 2 * lane));
        f1 = field_val(f.tlo1, f.thi1, field_addr(mt, n, 2 * lane + 1));
    }
}

// Reductions over lanes 0..15 (one DPP row); result valid in lane 15, broadcast with readlane.
__device__ __forceinline__ uint32_t row16_min_bcast(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ uint32_t row16_incl_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}

__device__ __forceinline__ double readlane_f64(double v, uint32_t src_lane) {  // src_lane uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t wave_shr1_u32(uint32_t v) {  // lane l gets lane l-1 (lane 0: itself)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ double wave_shr1_f64(double v) {
    const uint32_t lo = wave_shr1_u32((uint32_t)__double2loint(v));
    const uint32_t hi = wave_shr1_u32((uint32_t)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}

// ---------------------------------------------------------------------------
// Running top-k of ONE wave held in registers: RK rows of 64 entries, sorted best first, entry e
// in row e / 64 at lane e % 64.  Insert = ballot/popcount for the position, DPP wave shift,
// rows chained through lane 63 -> lane 0.  No LDS traffic.
// ---------------------------------------------------------------------------
template <int RK>
struct RegTopK {
    double score[RK];
    uint32_t doc[RK];
    uint32_t cnt;
    double kth_s;   // k-th entry, uniform copies (valid once cnt == k)
    uint32_t kth_d;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int r = 0; r < RK; ++r) {
            score[r] = 0.0;
            doc[r] = NONE32;
        }
        cnt = 0;
        kth_s = 0.0;
        kth_d = 0;
    }
    // offer one candidate per lane (`has` marks validity); all 64 lanes call
    __device__ __forceinline__ void offer(bool has, double sc, uint32_t d, uint32_t k, uint32_t lane) {
        for (;;) {
            const bool alive = has && (cnt < k || better(sc, d, kth_s, kth_d));
            const unsigned long long mask = __ballot(alive);
            if (!mask) break;
            const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1;
            const double cs = readlane_f64(sc, leader);
            const uint32_t cd = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)leader);
            if (lane == leader) has = false;
            uint32_t pos = 0;  // entries better than the candidate: a prefix of the list
#pragma unroll
            for (int r = 0; r < RK; ++r)
                pos += (uint32_t)__popcll(__ballot(r * 64 + lane < cnt && better(score[r], doc[r], cs, cd)));
            double carry_s = 0.0;
            uint32_t carry_d = NONE32;
#pragma unroll
            for (int r = 0; r < RK; ++r) {
                const double us = wave_shr1_f64(score[r]);
                const uint32_t ud = wave_shr1_u32(doc[r]);
                const double out_s = readlane_f64(score[r], 63);
                const uint32_t out_d = (uint32_t)__builtin_amdgcn_readlane((int)doc[r], 63);
                const uint32_t e = r * 64 + lane;
                if (e > pos) {
                    score[r] = lane == 0 ? carry_s : us;
                    doc[r] = lane == 0 ? carry_d : ud;
                } else if (e == pos) {
                    score[r] = cs;
                    doc[r] = cd;
                }
                carry_s = out_s;
                carry_d = out_d;
            }
            cnt = cnt < k ? cnt + 1 : k;
            if (cnt >= k) {
#pragma unroll
                for (int r = 0; r < RK; ++r)
                    if ((k - 1) / 64 == (uint32_t)r) {
                        kth_s = readlane_f64(score[r], (k - 1) & 63);
                        kth_d = (uint32_t)__builtin_amdgcn_readlane((int)doc[r], (int)((k - 1) & 63));
                    }
            }
        }
    }
};


// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// would wait for every global load in flight (the planner's metadata refills, the threshold
// poll); all hand-offs inside the tile loop go through LDS.
__device__ __forceinline__ uint32_t uni(uint32_t v) {  // value is wave-uniform: keep it in an SGPR
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint4 uni4(const uint4 v) {
    return make_uint4(uni(v.x), uni(v.y), uni(v.z), uni(v.w));
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef VBM25_PROFILE
#define PROF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define PROF_ADD(slot, a, b) prof[slot] += (b) - (a)
#else
#define PROF_T(var)
#define PROF_ADD(slot, a, b)
#endif

// ---------------------------------------------------------------------------
// Posting scan, tile formulation (scan_kernel): queries with CUR_T < terms <= CHAIN_MAX_TERMS, and every
// query of at most CHAIN_MAX_TERMS terms when k > REG_K or the cursor kernel is switched off.
//
// Per workgroup: CNW worker waves + one planner / merger wave + one joiner wave; C_BLOCKS block slots of
// staging in LDS (doc id, tf, fieldnorm per posting), split into per-term regions in key order, so that
// a posting's staging index orders postings by term.  Per doc-range tile [lo, hi), ONE LDS-only barrier:
//   plan    (planner, one tile ahead) hi = smallest min_doc of the first block that does not fit a term's
//           region; entries = newly admitted blocks + blocks still resident from earlier tiles
//           ("carried", decoded once per chunk); block metadata comes from an LDS ring
//   pass A  (workers, two entries each) decode (unless carried) from words fetched one tile earlier, stage,
//           mark every posting of [lo, hi) in two independently hashed seen / multi bitmap pairs
//   pass B  (workers) postings whose multi bit is clear under either hash are whole documents: dropped in
//           hot tiles (threshold above every token upper bound), else scored and filtered; the others go
//           to the tile's slow list
//   join    (joiner, one tile late) exact join of the slow list in registers, sums in key order
//   merge   (planner) running top-k in registers (k <= REG_K) or LDS; the k-th score is shared through LDS
//           and, across the chunks of a query, through a 64-bit atomicMax on the score bits
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    // DPP row shifts inside 16-lane rows, then row broadcasts across rows (gfx9 wave64)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return x;
}


template <int KMAX>
__global__ void __launch_bounds__(CWG, 6) scan_kernel(DevIndex ix, DevBatch bt) {
    constexpr int T = CHAIN_MAX_TERMS;
    constexpr int RING = 64;              // metadata ring entries (power-of-two ring per term)
    constexpr uint32_t PLANNER = CNW;     // waves 0..CNW-1 work: entries w and w + CNW of a tile
    constexpr uint32_t JOINER = CNW + 1;  // exact join of colliding postings, one tile late
    // staging: decoded postings of the resident blocks (needed again when a block is carried)
    __shared__ uint32_t st_doc[C_POSTINGS];
    __shared__ uint32_t st_tf[C_POSTINGS];
    __shared__ uint8_t st_fn[C_POSTINGS];
    // two independently hashed bitmap pairs per tile, three tiles in rotation: "some posting hit
    // this bit" / "a second posting hit it".  A posting is slow only if it collides under BOTH.
    __shared__ uint32_t bm_seen[3][2][BM_WORDS];
    __shared__ uint32_t bm_multi[3][2][BM_WORDS];
    __shared__ uint32_t sl_doc[2][SLOW_CAP];  // slow postings of a tile (copies)
    __shared__ double sl_p[2][SLOW_CAP];
    __shared__ uint16_t sl_idx[2][SLOW_CAP];
    __shared__ double jc_score[2][JC_CAP];    // documents produced by the join, for the merger
    __shared__ uint32_t jc_doc[2][JC_CAP];
    __shared__ double c_score[2][CAND_CAP];   // documents of the fast path (overflow: global spill)
    __shared__ uint32_t c_doc[2][CAND_CAP];
    __shared__ double s_s1[256];
    __shared__ TopK<(KMAX > REG_K ? KMAX : 1)> s_top;  // LDS list only for k > REG_K
    __shared__ uint4 s_ring[RING];
    __shared__ uint4 e_meta[2][C_BLOCKS];     // entries of a tile: new blocks first, then carried
    __shared__ uint2 e_aux[2][C_BLOCKS];      // {block index, staging base | term << 16}
    __shared__ double t_s0[T];
    // tile header {lo, hi, nent, nnew | done << 16}, per-tile counters, shared filter state
    __shared__ uint4 s_hdr[2];
    __shared__ uint32_t s_cand_cnt[2], sl_cnt[2], jc_cnt[2], s_abort;
    __shared__ unsigned long long s_theta;
    __shared__ double s_kth_score;
    __shared__ uint32_t s_top_cnt;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t k = bt.k;
    for (int i = tid; i < 256; i += CWG) s_s1[i] = ix.s1[i];
    for (int i = tid; i < 6 * BM_WORDS; i += CWG) {
        (&bm_seen[0][0][0])[i] = 0;
        (&bm_multi[0][0][0])[i] = 0;
    }
    // rarely used overflow areas in HBM, per workgroup: [cand | slow | join][2 bufs][C_POSTINGS][2 words]
    unsigned long long *spill_s = bt.spill + (size_t)blockIdx.x * 3 * 2 * C_POSTINGS * 2;
    unsigned long long *spill_l = spill_s + 2 * C_POSTINGS * 2;
    unsigned long long *spill_j = spill_l + 2 * C_POSTINGS * 2;

#ifdef VBM25_PROFILE
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif
    const uint32_t n_items = *bt.n_items;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const Item it = bt.items[item];
        if (it.m > (uint32_t)T || it.m < bt.chain_min_terms) continue;  // the other kernels'
        const uint32_t q = it.q, clo = it.doc_lo, chi = it.doc_hi;
        __syncthreads();
        if (tid == 0) s_abort = 0;

        if (wave == PLANNER) {
            // =====================================================================
            // Planner / merger wave: lane t owns term t.  Plans one tile ahead (block
            // metadata only) and owns the running top-k.
            // =====================================================================
            uint32_t p_rb = 0, p_re = 0, p_end = 0, p_q = 1, p_rmask = 0, p_roff = 0, p_base = 0,
                     p_slot = 0;  // p_slot = region slot of block p_rb (p_rb mod p_q, incremental)
            uint32_t m = 0;
            unsigned long long ub_bits = 0;  // bits of the largest single-posting score (+ margin)
            {
                const uint32_t qb = bt.q_off[q], qe = bt.q_off[q + 1];
                uint32_t term = NONE32;
                for (uint32_t p = qb; p < qe; ++p) {  // indexed terms in ascending key order
                    const uint32_t tt = bt.term_ids[p];
                    if (tt >= ix.n_terms) continue;  // search.rs:59-61
                    if (m == lane) term = tt;
                    ++m;
                }
                const bool act = lane < m;
                unsigned long long df = act ? ix.term_df[term] : 0ull;
                unsigned long long sum = df, frac = 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                if (act) {
                    const uint32_t b0 = ix.term_first_block[term], b1 = ix.term_first_block[term + 1];
                    uint32_t lo_b = b0, hi_b = b1;  // first block whose max_doc >= clo
                    while (lo_b < hi_b) {
                        const uint32_t mid = (lo_b + hi_b) >> 1;
                        if (ix.blk_max_doc[mid] < clo) lo_b = mid + 1; else hi_b = mid;
                    }
                    p_rb = p_re = lo_b;
                    p_end = b1;
                    p_q = (uint32_t)(((unsigned long long)(C_BLOCKS - m) * df) / sum) + 1;
                    frac = ((unsigned long long)(C_BLOCKS - m) * df) % sum;
                    t_s0[lane] = ix.term_s0[term];
                }
                {   // Cursor::new, search.rs:363: token_upper_bound = Cache::evaluate(token WAND pair)
                    double ub = 0.0;
                    if (act) {
                        const double wtf = (double)ix.term_wand_tf[term];
                        ub = (wtf * ix.term_s0[term]) / (wtf + ix.s1[ix.term_wand_fn[term]]);
                    }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) ub = fmax(ub, __shfl_xor(ub, o));
                    // margin: the pair maximises tf() at flush time; Cache::evaluate of another
                    // posting may round one ulp higher
                    ub_bits = (unsigned long long)__double_as_longlong(readlane_f64(ub, 0) * (1.0 + 1e-12));
                }
                {   // hand the block slots left over by the floor() to the largest remainders
                    const uint32_t used = row16_incl_sum(act ? p_q : 0u);
                    const uint32_t left = (uint32_t)C_BLOCKS - (uint32_t)__builtin_amdgcn_readlane((int)used, 15);
                    uint32_t rank = 0;
                    for (uint32_t t = 0; t < m; ++t) {
                        const unsigned long long ft = __shfl(frac, (int)t);
                        rank += (ft > frac || (ft == frac && t < lane)) ? 1u : 0u;
                    }
                    if (act && rank < left) p_q += 1;
                    uint32_t rs = 2;  // ring holds blocks [rb, rb + 2q]
                    while (rs < 2 * p_q + 1) rs <<= 1;
                    p_rmask = rs - 1;
                }
                const uint32_t xb = act ? 128 * p_q : 0, xr = act ? p_rmask + 1 : 0;
                p_base = row16_incl_sum(xb) - xb;
                p_roff = row16_incl_sum(xr) - xr;
                if (act) {  // initial fill of the metadata ring: blocks [rb, rb + 2q]
                    for (uint32_t i = 0; i <= 2 * p_q; ++i) {
                        const uint32_t j = p_rb + i;
                        if (j < p_end) s_ring[p_roff + (j & p_rmask)] = ix.blk_meta[j];
                    }
                }
                if (lane == 0) {
                    s_top.count = 0;
                    s_cand_cnt[0] = s_cand_cnt[1] = 0;
                    sl_cnt[0] = sl_cnt[1] = 0;
                    jc_cnt[0] = jc_cnt[1] = 0;
                    s_top_cnt = 0;
                    s_kth_score = 0.0;
                    s_theta = 0;
                }
            }
            const bool act = lane < m;
            uint32_t p_hi = clo;  // end of the tile planned last
            uint4 pf0 = make_uint4(0, 0, 0, 0), pf1 = pf0;
            uint32_t at0 = NONE32, at1 = NONE32;
            unsigned long long theta_next = 0;

            // plan the tile after [.., p_hi) into buffer nb (header + entries).  The loads it
            // starts are consumed by plan_finish(), after the next barrier.
            auto plan_start = [&](uint32_t nb) {
                const uint32_t hi_prev = p_hi;
                uint32_t nrb = p_rb;
                if (act) {  // 1. drop blocks that end before the previous tile's end
                    while (nrb < p_re && s_ring[p_roff + (nrb & p_rmask)].y < hi_prev) ++nrb;
                    p_slot += nrb - p_rb;
                    while (p_slot >= p_q) p_slot -= p_q;
                }
                at0 = at1 = NONE32;
                if (act) {  // 2. refill the ring towards [nrb, nrb + 2q] (used one tile later)
                    uint32_t j2 = p_rb + 2 * p_q + 1;
                    const uint32_t last = min(nrb + 2 * p_q, p_end - 1);
                    if (j2 <= last) {
                        pf0 = ix.blk_meta[j2];
                        at0 = p_roff + (j2 & p_rmask);
                        ++j2;
                    }
                    if (j2 <= last) {
                        pf1 = ix.blk_meta[j2];
                        at1 = p_roff + (j2 & p_rmask);
                        ++j2;
                    }
                    for (; j2 <= last; ++j2) s_ring[p_roff + (j2 & p_rmask)] = ix.blk_meta[j2];
                    p_rb = nrb;
                }
                theta_next = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // 3. tile range
                uint32_t lo_c = chi, hi_c = chi;
                if (act && p_rb < p_end) {
                    lo_c = max(hi_prev, s_ring[p_roff + (p_rb & p_rmask)].x);
                    if (p_rb + p_q < p_end) hi_c = s_ring[p_roff + ((p_rb + p_q) & p_rmask)].x;
                }
                const uint32_t lo_n = row16_min_bcast(lo_c), hi_n = min(chi, row16_min_bcast(hi_c));
                // 4. entries: newly admitted blocks first (they cost a decode), then the blocks
                //    still resident from earlier tiles
                uint32_t n_car = 0, n_new = 0, re_old = p_re;
                if (act && lo_n < chi) {
                    const uint32_t lim = min(p_rb + p_q, p_end);
                    re_old = max(p_re, p_rb);
                    uint32_t j = re_old;
                    while (j < lim && s_ring[p_roff + (j & p_rmask)].x < hi_n) ++j;
                    n_car = re_old - p_rb;
                    n_new = j - re_old;
                    p_re = j;
                }
                const uint32_t inc_n = row16_incl_sum(n_new), inc_c = row16_incl_sum(n_car);
                const uint32_t tot_new = (uint32_t)__builtin_amdgcn_readlane((int)inc_n, 15);
                const uint32_t tot_car = (uint32_t)__builtin_amdgcn_readlane((int)inc_c, 15);
                if (act) {
                    uint32_t slot = p_slot;
                    uint32_t e_c = tot_new + inc_c - n_car, e_n = inc_n - n_new;
                    for (uint32_t i = 0; i < n_car + n_new; ++i) {
                        const uint32_t j = p_rb + i;
                        const uint32_t e = i < n_car ? e_c++ : e_n++;
                        e_meta[nb][e] = s_ring[p_roff + (j & p_rmask)];
                        e_aux[nb][e] = make_uint2(j, (p_base + slot * 128) | (lane << 16));
                        if (++slot == p_q) slot = 0;
                    }
                }
                const bool fin = lo_n >= chi;
                // hot tile: the shared threshold already exceeds every single-posting score, so only
                // documents with two or more postings can still enter the top-k
                const bool hot = theta_next > ub_bits;
#ifdef VBM25_PROFILE
                prof[13] += hot ? 1 : 0;
#endif
                if (lane == 0)
                    s_hdr[nb] = make_uint4(lo_n, hi_n, tot_new + tot_car,
                                           tot_new | (fin ? 0x10000u : 0u) | (hot ? 0x20000u : 0u));
                p_hi = hi_n;
                return fin;
            };
            auto plan_finish = [&]() {
                if (at0 != NONE32) s_ring[at0] = pf0;
                if (at1 != NONE32) s_ring[at1] = pf1;
                if (lane == 0) s_theta = theta_next;
            };

            // running top-k: for k <= REG_K in registers (RegTopK), else a sorted list in LDS
            constexpr int RK = KMAX <= REG_K ? KMAX / 64 : 1;
            RegTopK<RK> rtop;
            rtop.init();
            auto offer1 = [&](bool has, double sc, uint32_t d) {
                if constexpr (KMAX <= REG_K) rtop.offer(has, sc, d, k, lane);
                else topk_offer<(KMAX > REG_K ? KMAX : 1)>(s_top, k, has, sc, d, lane);
            };
            auto offer_list = [&](const double *sc_arr, const uint32_t *d_arr, uint32_t cnt) {
                for (uint32_t base = 0; base < cnt; base += 64) {
                    const bool has = base + lane < cnt;
                    offer1(has, has ? sc_arr[base + lane] : 0.0, has ? d_arr[base + lane] : 0u);
                }
            };
            unsigned long long published = 0;
            auto publish = [&]() {  // new k-th entry -> candidate filters of this and other chunks
                uint32_t n_now;
                double ks = 0.0;
                if constexpr (KMAX <= REG_K) {
                    n_now = rtop.cnt;
                    ks = rtop.kth_s;
                } else {
                    n_now = s_top.count;
                    if (n_now >= k) {
                        ks = s_top.score[k - 1];
                    }
                }
                if (lane == 0) {
                    s_top_cnt = n_now;
                    if (n_now >= k) {
                        s_kth_score = ks;
                        const unsigned long long bits = (unsigned long long)__double_as_longlong(ks);
                        if (bits > published) {
                            atomicMax(&bt.theta[q], bits);
                            published = bits;
                        }
                    }
                }
            };
            // fast-path documents of tile buffer b (LDS part + global spill); also detects a slow
            // list that did not fit (-> abort the item, it is redone by scan_many_kernel)
            auto merge_cand = [&](uint32_t b) {
                const uint32_t cnt = uni(s_cand_cnt[b]);
                if (uni(sl_cnt[b]) > (uint32_t)SLOW_ABORT && lane == 0) s_abort = 1;
#ifdef VBM25_PROFILE
                {
                    const uint32_t ns = uni(sl_cnt[b]);
                    if (ns > prof[8]) prof[8] = ns;
                    prof[9] += ns;
                    prof[10] += ns > 64 ? 1 : 0;
                    prof[11] += cnt;
                    if (cnt > prof[12]) prof[12] = cnt;
                }
#endif
                if (!cnt) return;
                offer_list(c_score[b], c_doc[b], min(cnt, (uint32_t)CAND_CAP));
                for (uint32_t base = CAND_CAP; base < cnt; base += 64) {
                    const bool has = base + lane < cnt;
                    double sc = 0.0;
                    uint32_t d = 0;
                    if (has) {
                        const unsigned long long *sp = spill_s + ((size_t)b * C_POSTINGS + (base + lane - CAND_CAP)) * 2;
                        sc = __longlong_as_double((long long)__hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        d = (uint32_t)__hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    offer1(has, sc, d);
                }
                if (lane == 0) s_cand_cnt[b] = 0;
                publish();
            };
            // documents produced by the joiner for tile buffer b
            auto merge_jc = [&](uint32_t b) {
                const uint32_t jcnt = uni(jc_cnt[b]);
                if (!jcnt) return;
                offer_list(jc_score[b], jc_doc[b], min(jcnt, (uint32_t)JC_CAP));
                for (uint32_t base = JC_CAP; base < jcnt; base += 64) {
                    const bool has = base + lane < jcnt;
                    double sc = 0.0;
                    uint32_t d = 0;
                    if (has) {
                        const unsigned long long *sp = spill_j + ((size_t)b * C_POSTINGS + (base + lane - JC_CAP)) * 2;
                        sc = __longlong_as_double((long long)__hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        d = (uint32_t)__hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    offer1(has, sc, d);
                }
                if (lane == 0) jc_cnt[b] = 0;
                publish();
            };

            bool done = plan_start(0);
            plan_finish();
            lds_barrier();  // S
            // tile i: plan i+1, barrier X_i, then merge what is complete: fast path of tile i-1
            // (its pass B ended before X_i) and the join of tile i-2 (the joiner ran it between
            // X_{i-1} and X_i); both live in buffer (i-1) & 1 ... (i-2) & 1 respectively
            for (uint32_t par = 0;; par ^= 1) {
                if (done) break;
                const bool next_done = plan_start(par ^ 1);
                lds_barrier();  // X
                if (uni(s_abort)) break;
                plan_finish();
                merge_cand(par ^ 1);  // fast path of tile i-1: its pass B ended before X_i
                merge_jc(par);        // join of tile i-2: ran between X_{i-1} and X_i
#ifdef VBM25_PROFILE
                prof[7] += 1;
#endif
                done = next_done;
            }
            __syncthreads();  // E1: every wave left the tile loop; the joiner flushes its list
            __syncthreads();  // E2
            const bool failed = uni(s_abort) != 0 || uni(sl_cnt[0]) > (uint32_t)SLOW_ABORT || uni(sl_cnt[1]) > (uint32_t)SLOW_ABORT;
            merge_cand(0);
            merge_cand(1);
            merge_jc(0);
            merge_jc(1);
            if (lane == 0) bt.item_failed[item] = failed ? 1u : 0u;
#ifdef VBM25_PROFILE
            prof[6] += failed ? 1 : 0;
#endif
            if constexpr (KMAX <= REG_K) {  // chunk result straight from the registers
#pragma unroll
                for (int r = 0; r < RK; ++r) {
                    const uint32_t e = r * 64 + lane;
                    if (e < rtop.cnt) {
                        bt.res_score[(size_t)item * k + e] = rtop.score[r];
                        bt.res_doc[(size_t)item * k + e] = rtop.doc[r];
                    }
                }
                if (lane == 0) bt.res_cnt[item] = rtop.cnt;
            }
        } else if (wave == JOINER) {
            // =====================================================================
            // Joiner wave: exact join of the postings that collided under both hashes.  Each
            // lane holds one of them and meets all the others through readlane (no LDS traffic);
            // group leader = smallest staging index = first key.  The list of tile i is complete
            // at barrier X_{i+1} and is joined before X_{i+2}.
            // =====================================================================
            // item e of tile buffer b: the first SLOW_CAP live in LDS, the rest in the global spill
            auto item_at = [&](uint32_t b, uint32_t e, uint32_t &d, uint32_t &idx, double &p) {
                if (e < (uint32_t)SLOW_CAP) {
                    d = sl_doc[b][e];
                    idx = sl_idx[b][e];
                    p = sl_p[b][e];
                } else {
                    const unsigned long long *sp = spill_l + ((size_t)b * C_POSTINGS + (e - SLOW_CAP)) * 2;
                    p = __longlong_as_double((long long)__hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    const unsigned long long w = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    d = (uint32_t)w;
                    idx = (uint32_t)(w >> 32);
                }
            };
            auto emit = [&](uint32_t b, bool lead, double score, uint32_t d) {
                // pre-filter on the score alone, strictly: the merger publishes {count, score, doc} with
                // separate stores while this wave runs, and a torn (new score, old doc) pair must not
                // drop a document that ties the k-th score; the merger applies the exact rule
                if (lead && !(s_top_cnt >= k && score < s_kth_score)) {
                    const uint32_t at = atomicAdd(&jc_cnt[b], 1u);
                    if (at < (uint32_t)JC_CAP) {
                        jc_score[b][at] = score;
                        jc_doc[b][at] = d;
                    } else {
                        unsigned long long *sp = spill_j + ((size_t)b * C_POSTINGS + (at - JC_CAP)) * 2;
                        __hip_atomic_store(sp, (unsigned long long)__double_as_longlong(score), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(sp + 1, (unsigned long long)d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
            };
            auto join = [&](uint32_t b) {
                const uint32_t n = uni(sl_cnt[b]);
                if (n == 0 || n > (uint32_t)SLOW_ABORT) return;  // overflow: the planner aborts the item
                if (n <= 64) {
                    const bool v = lane < n;
                    const uint32_t jd = v ? sl_doc[b][lane] : NONE32;
                    const uint32_t ji = v ? (uint32_t)sl_idx[b][lane] : NONE32;
                    const double jp = v ? sl_p[b][lane] : 0.0;
                    uint32_t same = 0, minidx = ji, mate = 0;
                    for (uint32_t j = 0; j < n; ++j) {
                        const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)jd, (int)j);
                        const uint32_t ij = (uint32_t)__builtin_amdgcn_readlane((int)ji, (int)j);
                        const bool hit = dj == jd && ij != ji;
                        same += hit ? 1u : 0u;
                        mate = hit ? j : mate;
                        minidx = hit ? min(minidx, ij) : minidx;
                    }
                    const bool lead = v && minidx == ji;
                    double score = jp;
                    if (__ballot(lead && same >= 1)) {
                        const double op = __shfl(jp, (int)mate);
                        if (lead && same == 1) score = jp + op;  // two addends commute
                    }
                    if (__ballot(lead && same >= 2)) {
                        // three or more addends: ascending staging index = key order, one pass
                        // over the list per addend
                        double acc = 0.0;
                        int last = -1;
                        const bool l3 = lead && same >= 2;
                        for (;;) {
                            uint32_t best = NONE32;
                            double bp = 0.0;
                            for (uint32_t j = 0; j < n; ++j) {
                                const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)jd, (int)j);
                                const uint32_t ij = (uint32_t)__builtin_amdgcn_readlane((int)ji, (int)j);
                                const double pj = readlane_f64(jp, j);
                                if (l3 && dj == jd && (int)ij > last && ij < best) {
                                    best = ij;
                                    bp = pj;
                                }
                            }
                            if (!__ballot(best != NONE32)) break;
                            if (best != NONE32) {
                                acc += bp;
                                last = (int)best;
                            }
                        }
                        if (l3) score = acc;
                    }
                    emit(b, lead, score, jd);
                } else {
                    // rare: a long list.  Same join, every lane owns one item per round and reads
                    // all the others (LDS / spill broadcast reads).
                    for (uint32_t base = 0; base < n; base += 64) {
                        const bool v = base + lane < n;
                        uint32_t jd = NONE32, ji = NONE32;
                        double jp = 0.0;
                        if (v) item_at(b, base + lane, jd, ji, jp);
                        uint32_t same = 0, minidx = ji;
                        for (uint32_t j = 0; j < n; ++j) {
                            uint32_t dj, ij;
                            double pj;
                            item_at(b, j, dj, ij, pj);
                            const bool hit = dj == jd && ij != ji;
                            same += hit ? 1u : 0u;
                            minidx = hit ? min(minidx, ij) : minidx;
                        }
                        const bool lead = v && minidx == ji;
                        double score = jp;
                        if (__ballot(lead && same >= 1)) {  // ordered sum over the group
                            double acc = 0.0;
                            int last = -1;
                            const bool l2 = lead && same >= 1;
                            for (;;) {
                                uint32_t best = NONE32;
                                double bp = 0.0;
                                for (uint32_t j = 0; j < n; ++j) {
                                    uint32_t dj, ij;
                                    double pj;
                                    item_at(b, j, dj, ij, pj);
                                    if (l2 && dj == jd && (int)ij > last && ij < best) {
                                        best = ij;
                                        bp = pj;
                                    }
                                }
                                if (!__ballot(best != NONE32)) break;
                                if (best != NONE32) {
                                    acc += bp;
                                    last = (int)best;
                                }
                            }
                            if (l2) score = acc;
                        }
                        emit(b, lead, score, jd);
                    }
                }
                if (lane == 0) sl_cnt[b] = 0;
            };
            lds_barrier();  // S
            for (uint32_t par = 0;; par ^= 1) {
                if (uni(s_hdr[par].w) & 0x10000u) break;
                lds_barrier();  // X
                if (uni(s_abort)) break;
                join(par ^ 1);  // the previous tile's list
            }
            __syncthreads();  // E1
            join(0);
            join(1);
            __syncthreads();  // E2
        } else {
            // =====================================================================
            // Worker waves: entries w and w + CNW of every tile; ONE barrier per tile
            // =====================================================================
            BlockFetch fetch[2];
            uint4 ent_m[2];   // this tile's entries (wave-uniform)
            uint2 ent_a[2];
            bool fetched = false;
            lds_barrier();  // S
            uint4 hdr = uni4(s_hdr[0]);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t e = wave + r * CNW;
                ent_m[r] = uni4(e_meta[0][e < (uint32_t)C_BLOCKS ? e : 0]);
                const uint2 a = e_aux[0][e < (uint32_t)C_BLOCKS ? e : 0];
                ent_a[r] = make_uint2(uni(a.x), uni(a.y));
            }
            unsigned long long theta = 0;
            uint32_t ntop = 0;
            double kscore = 0.0;
            uint32_t tile = 0;
            for (uint32_t par = 0;; par ^= 1, ++tile) {
                if (hdr.w & 0x10000u) break;
                const uint32_t lo = hdr.x, hi = hdr.y, nent = hdr.z, nnew = hdr.w & 0xffffu;
                const uint32_t bbuf = tile % 3;

                // ---- pass A.1: decode this wave's new blocks into staging; fetch carried ones
                uint32_t dd[4], tt[4], fnp[2];  // doc ids, term frequencies, packed fieldnorm pairs
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t e = wave + r * CNW;
                    dd[2 * r] = dd[2 * r + 1] = NONE32;
                    tt[2 * r] = tt[2 * r + 1] = 0;
                    fnp[r] = 0;
                    const uint32_t i0 = (ent_a[r].y & 0xffffu) + 2 * lane;
                    if (e >= nent) continue;
                    if (e >= nnew) {  // carried over from an earlier tile: already staged
                        const uint2 v = *reinterpret_cast<const uint2 *>(&st_doc[i0]);
                        const uint2 w = *reinterpret_cast<const uint2 *>(&st_tf[i0]);
                        dd[2 * r] = v.x;
                        dd[2 * r + 1] = v.y;
                        tt[2 * r] = w.x;
                        tt[2 * r + 1] = w.y;
                        fnp[r] = *reinterpret_cast<const uint16_t *>(&st_fn[i0]);
                        continue;
                    }
                    const uint4 bm = ent_m[r];
                    const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff;
                    if (!fetched) block_fetch(ix, bm, ent_a[r].x, lane, fetch[r]);
                    const BlockFetch &f = fetch[r];
                    uint32_t v0, v1, f0, f1;
                    block_fields(bm, lane, f, v0, v1, f0, f1);
                    uint32_t d0 = v0, d1 = v1;
                    const uint32_t width = md & 127u;
                    if (!((md >> 7) ? (width == 4) : (width == 32))) {  // d1 deltas from min_doc
                        const uint32_t own = v0 + v1;
                        const uint32_t incl = wave_incl_scan_u32(own);
                        d0 = bm.x + (incl - own) + v0;
                        d1 = d0 + v1;
                    }
                    if (2 * lane >= n) d0 = NONE32;
                    if (2 * lane + 1 >= n) d1 = NONE32;
                    *reinterpret_cast<uint2 *>(&st_doc[i0]) = make_uint2(d0, d1);
                    *reinterpret_cast<uint2 *>(&st_tf[i0]) = make_uint2(f0, f1);
                    *reinterpret_cast<uint16_t *>(&st_fn[i0]) = (uint16_t)f.fn;
                    dd[2 * r] = d0;
                    dd[2 * r + 1] = d1;
                    tt[2 * r] = f0;
                    tt[2 * r + 1] = f1;
                    fnp[r] = f.fn;
                }
                // ---- pass A.2: mark every posting of [lo, hi) in the hashed bitmaps.  A bit that
                // was already set means "another posting may belong to the same document".
                uint32_t inr = 0;  // bit x: posting x is inside [lo, hi)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const uint32_t d = dd[x];
                    if (d >= lo && d < hi) {  // NONE32 never is
                        inr |= 1u << x;
                        const uint32_t h = d & ((1u << BM_BITS_LOG2) - 1u);  // ids of a tile lie in a narrow range
                        const uint32_t g = (__umul24(d >> BM_BITS_LOG2, 97u) + d) & ((1u << BM_BITS_LOG2) - 1u);  // never equal for two ids that share h
                        const uint32_t hb = 1u << (h & 31), gb = 1u << (g & 31);
                        const uint32_t o1 = atomicOr(&bm_seen[bbuf][0][h >> 5], hb);
                        const uint32_t o2 = atomicOr(&bm_seen[bbuf][1][g >> 5], gb);
                        if (o1 & hb) atomicOr(&bm_multi[bbuf][0][h >> 5], hb);
                        if (o2 & gb) atomicOr(&bm_multi[bbuf][1][g >> 5], gb);
                    }
                }
                lds_barrier();  // X: all marks of this tile are in; everybody finished tile - 1
                if (uni(s_abort)) break;

                // ---- next tile: header, this wave's entries, filter state -- one LDS round trip;
                // then the loads of its new blocks
                const uint4 nh = uni4(s_hdr[par ^ 1]);
                uint4 nm[2];
                uint2 na[2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const uint32_t e = wave + r * CNW;
                    nm[r] = uni4(e_meta[par ^ 1][e < (uint32_t)C_BLOCKS ? e : 0]);
                    const uint2 a2 = e_aux[par ^ 1][e < (uint32_t)C_BLOCKS ? e : 0];
                    na[r] = make_uint2(uni(a2.x), uni(a2.y));
                }
                theta = s_theta;
                ntop = s_top_cnt;
                kscore = s_kth_score;
                // the bitmaps of the previous tile are free now (next used two tiles from here)
                {
                    const uint32_t wb = (tile + 2) % 3;
                    for (int i = tid; i < 2 * BM_WORDS / 4; i += CNW * 64) {
                        reinterpret_cast<uint4 *>(&bm_seen[wb][0][0])[i] = make_uint4(0, 0, 0, 0);
                        reinterpret_cast<uint4 *>(&bm_multi[wb][0][0])[i] = make_uint4(0, 0, 0, 0);
                    }
                }
                // ---- pass B: a document whose bit nobody else hit has a single posting: its
                // partial score IS its score.  In a hot tile such a document cannot reach the top-k
                // and no score is computed at all.  The others go to the joiner (with their score).
                const bool hot = (hdr.w & 0x20000u) != 0;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    if (!(inr & (1u << x))) continue;
                    const uint32_t d = dd[x];
                    const uint32_t h = d & ((1u << BM_BITS_LOG2) - 1u);  // ids of a tile lie in a narrow range
                    const uint32_t g = (__umul24(d >> BM_BITS_LOG2, 97u) + d) & ((1u << BM_BITS_LOG2) - 1u);  // never equal for two ids that share h
                    const bool single = !((bm_multi[bbuf][0][h >> 5] >> (h & 31)) & (bm_multi[bbuf][1][g >> 5] >> (g & 31)) & 1u);
                    if (single && hot) continue;
                    // Cache::evaluate, bm25.rs:355-358
                    const double tf = (double)tt[x];
                    const double p = (tf * t_s0[ent_a[x >> 1].y >> 16]) / (tf + s_s1[(fnp[x >> 1] >> (8 * (x & 1))) & 0xff]);
                    if (single) {
                        if ((unsigned long long)__double_as_longlong(p) < theta) continue;
                        // score alone, strictly (see the joiner's emit): ties go to the merger
                        if (ntop >= k && p < kscore) continue;
                        const uint32_t at = atomicAdd(&s_cand_cnt[par], 1u);
                        if (at < (uint32_t)CAND_CAP) {
                            c_score[par][at] = p;
                            c_doc[par][at] = d;
                        } else {  // cold tiles: more candidates than the LDS buffer holds
                            unsigned long long *sp = spill_s + ((size_t)par * C_POSTINGS + (at - CAND_CAP)) * 2;
                            __hip_atomic_store(sp, (unsigned long long)__double_as_longlong(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(sp + 1, (unsigned long long)d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // landed before the next barrier
                        }
                    } else {
                        const uint32_t idx = (ent_a[x >> 1].y & 0xffffu) + 2 * lane + (x & 1);  // staging index
                        const uint32_t at = atomicAdd(&sl_cnt[par], 1u);
                        if (at < (uint32_t)SLOW_CAP) {
                            sl_doc[par][at] = d;
                            sl_p[par][at] = p;
                            sl_idx[par][at] = (uint16_t)idx;
                        } else if (at < (uint32_t)SLOW_ABORT) {
                            unsigned long long *sp = spill_l + ((size_t)par * C_POSTINGS + (at - SLOW_CAP)) * 2;
                            __hip_atomic_store(sp, (unsigned long long)__double_as_longlong(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(sp + 1, (unsigned long long)d | (unsigned long long)idx << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                    }
                }
                // ---- loads of the next tile's new blocks (consumed after the next barrier)
                fetched = false;
                if (!(nh.w & 0x10000u)) {
                    const uint32_t nn = nh.w & 0xffffu;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const uint32_t e = wave + r * CNW;
                        if (e < nn) block_fetch(ix, nm[r], na[r].x, lane, fetch[r]);
                    }
                    fetched = true;
                }
                // roll over to the next tile
                hdr = nh;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    ent_m[r] = nm[r];
                    ent_a[r] = na[r];
                }
#ifdef VBM25_PROFILE
                prof[7] += nent;
#endif
            }
            __syncthreads();  // E1
            __syncthreads();  // E2
        }

        __syncthreads();
        if constexpr (KMAX > REG_K) {
            const uint32_t n = s_top.count;
            for (uint32_t i = tid; i < n; i += CWG) {
                bt.res_score[(size_t)item * k + i] = s_top.score[i];
                bt.res_doc[(size_t)item * k + i] = s_top.doc[i];
            }
            if (tid == 0) bt.res_cnt[item] = n;
        }
    }
#ifdef VBM25_PROFILE
    if (bt.prof && lane == 0 && (wave == 0 || wave == PLANNER)) {
        unsigned long long *o = bt.prof + (size_t)blockIdx.x * 33 + (wave == 0 ? 0 : 16);
        for (int i = 0; i < 16; ++i) o[i] = prof[i];
        if (wave == 0) bt.prof[(size_t)blockIdx.x * 33 + 32] = __builtin_readcyclecounter() - prof_t0;
    }
#endif
}

#include "scan_cursor.h"

// ---------------------------------------------------------------------------
// Merge of per-chunk lists -> hits
// ---------------------------------------------------------------------------
template <int KMAX>
__global__ void __launch_bounds__(64) merge_kernel(DevIndex ix, DevBatch bt) {
    __shared__ TopK<(KMAX > REG_K ? KMAX : 1)> s_top;
    constexpr int RK = KMAX <= REG_K ? KMAX / 64 : 1;
    RegTopK<RK> rtop;
    rtop.init();
    const uint32_t q = blockIdx.x, lane = threadIdx.x, k = bt.k;
    if (lane == 0) s_top.count = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t i0 = bt.q_item_base[q], i1 = bt.q_item_base[q + 1];
    for (uint32_t item = i0; item < i1; ++item) {
        const uint32_t cnt = uni(bt.res_cnt[item]);
        for (uint32_t base = 0; base < cnt; base += 64) {
            const bool has = base + lane < cnt;
            double sc = 0;
            uint32_t d = 0;
            if (has) {
                sc = bt.res_score[(size_t)item * k + base + lane];
                d = bt.res_doc[(size_t)item * k + base + lane];
            }
            if constexpr (KMAX <= REG_K) rtop.offer(has, sc, d, k, lane);
            else topk_offer<(KMAX > REG_K ? KMAX : 1)>(s_top, k, has, sc, d, lane);
        }
    }
    __builtin_amdgcn_wave_barrier();
    auto emit = [&](uint32_t i, double sc, uint32_t d) {
        // 24-byte record written as three 64-bit words so that padding bytes are zero
        const uint16_t *pl = ix.doc_payload + 3ull * d;
        unsigned long long *out = reinterpret_cast<unsigned long long *>(bt.hits + (size_t)q * k + i);
        out[0] = (unsigned long long)__double_as_longlong(sc);
        out[1] = (unsigned long long)d | (unsigned long long)pl[0] << 32 | (unsigned long long)pl[1] << 48;
        out[2] = (unsigned long long)pl[2];
    };
    uint32_t n;
    if constexpr (KMAX <= REG_K) {
        n = rtop.cnt;
#pragma unroll
        for (int r = 0; r < RK; ++r)
            if (r * 64 + lane < n) emit(r * 64 + lane, rtop.score[r], rtop.doc[r]);
    } else {
        n = s_top.count;
        for (uint32_t i = lane; i < n; i += 64) emit(i, s_top.score[i], s_top.doc[i]);
    }
    if (lane == 0) bt.n_hits[q] = n;
}

// ---------------------------------------------------------------------------
// Host objects
// ---------------------------------------------------------------------------
struct DeviceBuffer {
    void *p = nullptr;
    size_t bytes = 0;
    ~DeviceBuffer() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n) {
        bytes = n;
        HIP_TRY(hipMalloc(&p, n ? n : 16));
        return VBM25_OK;
    }
    int upload(const void *src, size_t n) {
        if (int rc = alloc(n)) return rc;
        if (n) HIP_TRY(hipMemcpy(p, src, n, hipMemcpyHostToDevice));
        return VBM25_OK;
    }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

}  // namespace vbm25

using namespace vbm25;

struct vbm25_index {
    int device = 0;
    DevIndex dev{};
    uint32_t n_docs = 0, n_terms = 0, n_blocks = 0;
    std::vector<uint8_t> term_key;  // host copy for vbm25_lookup_terms
    std::vector<uint32_t> term_df_host;  // host copy for query routing
    vbm25_batch *scratch = nullptr;      // batch object re-used by vbm25_search_batch
    DeviceBuffer term_wand_tf, term_wand_fn, term_df, term_first_block, term_s0, blk_min_doc, blk_max_doc, blk_meta, blk_ub, blob,
        post_fn, doc_payload, s1;
    uint64_t device_bytes = 0;
};

struct vbm25_batch {
    vbm25_index *index = nullptr;
    uint32_t max_queries = 0, max_terms = 0, k = 0, nq = 0, max_items = 0;
    DeviceBuffer term_ids, q_off, items, n_items, q_item_base, theta, res_score, res_doc, res_cnt,
        hits, n_hits, error_flag, prof, q_dense, spill, item_failed, work_ctr, hist;
    bool timing = false;
    bool has_many_terms = false;  // some query has more than CHAIN_MAX_TERMS indexed terms
    bool use_cursor = false;      // k <= REG_K: queries with at most CUR_T terms take scan_cursor_kernel
    bool has_mid_terms = false;   // some sparse query has CUR_T < terms <= CHAIN_MAX_TERMS
    uint32_t cur_mt = 1;          // most indexed terms among the cursor kernel's queries
    uint32_t target_items = TARGET_ITEMS;
    uint32_t min_chunk = MIN_CHUNK_POSTINGS;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    ~vbm25_batch() {
        for (auto &e : events) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    }
};

namespace {

int check_desc(const vbm25_index_desc *d) {
    if (!d) return set_error(VBM25_ERR_INVALID, "desc is NULL");
    if (!d->n_docs) return set_error(VBM25_ERR_INVALID, "index without documents");
    if (!(d->k1 >= 1.2 && d->k1 <= 2.0) || !(d->b >= 0.0 && d->b <= 1.0))
        return set_error(VBM25_ERR_INVALID, "k1 must be in [1.2, 2] and b in [0, 1]");
    if (d->n_terms && (!d->term_key || !d->term_df || !d->term_first_block || !d->term_wand_tf || !d->term_wand_fn))
        return set_error(VBM25_ERR_INVALID, "term arrays missing");
    if (d->n_blocks && (!d->blk_min_doc || !d->blk_max_doc || !d->blk_n || !d->blk_meta_doc ||
                        !d->blk_meta_tf || !d->blk_off8 || !d->blob))
        return set_error(VBM25_ERR_INVALID, "block arrays missing");
    if (!d->doc_fieldnorm || !d->doc_payload)
        return set_error(VBM25_ERR_INVALID, "document arrays missing");
    if (d->n_terms) {
        if (d->term_first_block[0] != 0 || d->term_first_block[d->n_terms] != d->n_blocks)
            return set_error(VBM25_ERR_CORRUPT, "term_first_block does not cover the blocks");
        for (uint32_t t = 0; t < d->n_terms; ++t) {
            uint32_t nb = d->term_first_block[t + 1] - d->term_first_block[t];
            if (d->term_first_block[t + 1] < d->term_first_block[t] ||
                nb != (d->term_df[t] + 127) / 128 || d->term_df[t] == 0 || d->term_df[t] > d->n_docs)
                return set_error(VBM25_ERR_CORRUPT, "term %u: df / block count mismatch", t);
            if (t && std::memcmp(d->term_key + 16ull * (t - 1), d->term_key + 16ull * t, 16) >= 0)
                return set_error(VBM25_ERR_CORRUPT, "term keys not strictly ascending at %u", t);
        }
    } else if (d->n_blocks) {
        return set_error(VBM25_ERR_CORRUPT, "blocks without terms");
    }
    for (uint32_t t = 0; t < d->n_terms; ++t) {
        uint32_t b0 = d->term_first_block[t], b1 = d->term_first_block[t + 1];
        uint64_t cnt = 0;
        for (uint32_t j = b0; j < b1; ++j) {
            const uint32_t n = d->blk_n[j];
            const uint8_t md = d->blk_meta_doc[j], mt = d->blk_meta_tf[j];
            if (n < 1 || n > 128 || (j + 1 < b1 && n != 128))
                return set_error(VBM25_ERR_CORRUPT, "block %u: bad posting count %u", j, n);
            const bool full = n == 128;
            for (uint8_t mm : {md, mt}) {
                const uint32_t w = mm & 127;
                if (full ? ((mm >> 7) != 0 || w > 32) : ((mm >> 7) != 1 || w < 1 || w > 4))
                    return set_error(VBM25_ERR_CORRUPT, "block %u: bad codec metadata 0x%02x", j, mm);
            }
            const uint32_t ld = (md >> 7) ? (md & 127u) * n : 16u * (md & 127u);
            const uint32_t lt = (mt >> 7) ? (mt & 127u) * n : 16u * (mt & 127u);
            const uint64_t need = ((ld + 7) / 8) + ((lt + 7) / 8);
            if (d->blk_off8[j + 1] < d->blk_off8[j] || d->blk_off8[j + 1] - d->blk_off8[j] != need)
                return set_error(VBM25_ERR_CORRUPT, "block %u: body length mismatch", j);
            if (d->blk_min_doc[j] > d->blk_max_doc[j] || d->blk_max_doc[j] >= d->n_docs ||
                (j > b0 && d->blk_min_doc[j] <= d->blk_max_doc[j - 1]))
                return set_error(VBM25_ERR_CORRUPT, "block %u: document range out of order", j);
            cnt += n;
        }
        if (cnt != d->term_df[t]) return set_error(VBM25_ERR_CORRUPT, "term %u: df mismatch", t);
    }
    if (d->n_blocks && 8ull * d->blk_off8[d->n_blocks] > d->blob_bytes)
        return set_error(VBM25_ERR_CORRUPT, "blob shorter than the block offsets");
    return VBM25_OK;
}

int use_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return VBM25_OK;
}

template <class F>
int dispatch_k(uint32_t k, F &&f) {
    if (k <= 64) return f(std::integral_constant<int, 64>());
    if (k <= 128) return f(std::integral_constant<int, 128>());
    if (k <= 256) return f(std::integral_constant<int, 256>());
    return f(std::integral_constant<int, 1024>());
}

}  // namespace

extern "C" {

const char *vbm25_last_error(void) { return g_error; }
const char *vbm25_version(void) { return "vbm25-mi355x 0.1 (gfx950)"; }

static int vbm25_index_create_impl(const vbm25_index_desc *d, int device, vbm25_index **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (int rc = check_desc(d)) return rc;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return set_error(VBM25_ERR_DEVICE, "no HIP device: the MI355X path has no CPU fallback");
    if (device < 0 || device >= n_dev)
        return set_error(VBM25_ERR_INVALID, "device %d out of range (%d devices)", device, n_dev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!std::strstr(prop.gcnArchName, "gfx950"))
        return set_error(VBM25_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only",
                         device, prop.gcnArchName);
    if (int rc = use_device(device)) return rc;

    auto ix = std::make_unique<vbm25_index>();
    ix->device = device;
    ix->n_docs = d->n_docs;
    ix->n_terms = d->n_terms;
    ix->n_blocks = d->n_blocks;
    ix->term_key.assign(d->term_key, d->term_key + 16ull * d->n_terms);
    ix->term_df_host.assign(d->term_df, d->term_df + d->n_terms);

    std::vector<double> s0(d->n_terms);
    for (uint32_t t = 0; t < d->n_terms; ++t) s0[t] = bm25_s0(d->n_docs, d->term_df[t], d->k1);
    double s1[256];
    bm25_tables(d->n_docs, d->sum_len, d->k1, d->b, s1);
    std::vector<uint4> meta(d->n_blocks);
    for (uint32_t j = 0; j < d->n_blocks; ++j) {
        meta[j].x = d->blk_min_doc[j];
        meta[j].y = d->blk_max_doc[j];
        meta[j].z = d->blk_off8[j];
        meta[j].w = uint32_t(d->blk_n[j]) | uint32_t(d->blk_meta_doc[j]) << 8 |
                    uint32_t(d->blk_meta_tf[j]) << 16 | uint32_t(d->blk_wand_fn ? d->blk_wand_fn[j] : 0) << 24;
    }
    // block upper bounds (search.rs:377-380 evaluates the block WAND pair per visited block; here once)
    std::vector<double> blk_ub(d->n_blocks);
    for (uint32_t t = 0; t < d->n_terms; ++t) {
        const double wtf = double(d->term_wand_tf[t]);
        const double tub = (wtf * s0[t]) / (wtf + s1[d->term_wand_fn[t]]);
        for (uint32_t j = d->term_first_block[t]; j < d->term_first_block[t + 1]; ++j) {
            double ub = tub;
            if (d->blk_wand_fn && d->blk_wand_tf) {
                const double tf = double(d->blk_wand_tf[j]);
                ub = (tf * s0[t]) / (tf + s1[d->blk_wand_fn[j]]);
            }
            blk_ub[j] = ub * (1.0 + 1e-12);  // margin: another posting's evaluate may round one ulp higher
        }
    }
    DeviceBuffer fieldnorm, err;
    int rc = 0;
    const size_t blob_alloc = ((size_t(d->blob_bytes) + 15) & ~size_t(15)) + 64;  // slack for word reads
    if ((rc = ix->term_df.upload(d->term_df, 4ull * d->n_terms)) ||
        (rc = ix->term_first_block.upload(d->term_first_block, 4ull * (d->n_terms + 1))) ||
        (rc = ix->term_s0.upload(s0.data(), 8ull * d->n_terms)) ||
        (rc = ix->term_wand_tf.upload(d->term_wand_tf, 4ull * d->n_terms)) ||
        (rc = ix->term_wand_fn.upload(d->term_wand_fn, d->n_terms)) ||
        (rc = ix->blk_min_doc.upload(d->blk_min_doc, 4ull * d->n_blocks)) ||
        (rc = ix->blk_max_doc.upload(d->blk_max_doc, 4ull * d->n_blocks)) ||
        (rc = ix->blk_meta.upload(meta.data(), 16ull * d->n_blocks)) ||
        (rc = ix->blk_ub.upload(blk_ub.data(), 8ull * d->n_blocks)) ||
        (rc = ix->blob.alloc(blob_alloc)) ||
        (rc = ix->post_fn.alloc(128ull * d->n_blocks)) ||
        (rc = ix->doc_payload.upload(d->doc_payload, 6ull * d->n_docs)) ||
        (rc = ix->s1.upload(s1, sizeof s1)) ||
        (rc = fieldnorm.upload(d->doc_fieldnorm, d->n_docs)) || (rc = err.alloc(4)))
        return rc;
    HIP_TRY(hipMemset(err.p, 0, 4));
    HIP_TRY(hipMemset(ix->blob.p, 0, blob_alloc));
    if (d->blob_bytes) HIP_TRY(hipMemcpy(ix->blob.p, d->blob, d->blob_bytes, hipMemcpyHostToDevice));
    if (d->n_blocks) {
        const uint32_t grid = (d->n_blocks + 3) / 4;
        PostFnArgs pa{};
        pa.n_blocks = d->n_blocks;
        pa.n_docs = d->n_docs;
        pa.n_terms = d->n_terms;
        pa.blk_meta = ix->blk_meta.as<uint4>();
        pa.blob = ix->blob.as<uint8_t>();
        pa.doc_fieldnorm = fieldnorm.as<uint8_t>();
        pa.post_fn = ix->post_fn.as<uint8_t>();
        pa.error_flag = err.as<uint32_t>();
        pa.term_first_block = ix->term_first_block.as<uint32_t>();
        pa.term_wand_tf = ix->term_wand_tf.as<uint32_t>();
        pa.term_wand_fn = ix->term_wand_fn.as<uint8_t>();
        pa.term_s0 = ix->term_s0.as<double>();
        pa.s1 = ix->s1.as<double>();
        pa.blk_ub = ix->blk_ub.as<double>();
        post_fn_kernel<<<grid, 256>>>(pa);
        HIP_TRY(hipGetLastError());
    }
    uint32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, err.p, 4, hipMemcpyDeviceToHost));
    if (flag & 1u)
        return set_error(VBM25_ERR_CORRUPT,
                         "posting blocks do not decode to strictly increasing ids within "
                         "[min_doc, max_doc] below n_docs");
    if (flag & 2u)
        return set_error(VBM25_ERR_CORRUPT,
                         "a posting scores above its block's / token's WAND pair "
                         "(search.rs:363,377-380 prune with those bounds)");
    ix->dev.n_docs = d->n_docs;
    ix->dev.n_terms = d->n_terms;
    ix->dev.n_blocks = d->n_blocks;
    ix->dev.term_df = ix->term_df.as<uint32_t>();
    ix->dev.term_first_block = ix->term_first_block.as<uint32_t>();
    ix->dev.term_s0 = ix->term_s0.as<double>();
    ix->dev.term_wand_tf = ix->term_wand_tf.as<uint32_t>();
    ix->dev.term_wand_fn = ix->term_wand_fn.as<uint8_t>();
    ix->dev.blk_min_doc = ix->blk_min_doc.as<uint32_t>();
    ix->dev.blk_max_doc = ix->blk_max_doc.as<uint32_t>();
    ix->dev.blk_meta = ix->blk_meta.as<uint4>();
    ix->dev.blk_ub = ix->blk_ub.as<double>();
    ix->dev.blob = ix->blob.as<uint8_t>();
    ix->dev.post_fn = ix->post_fn.as<uint8_t>();
    ix->dev.doc_payload = ix->doc_payload.as<uint16_t>();
    ix->dev.s1 = ix->s1.as<double>();
    for (const DeviceBuffer *b : {&ix->term_df, &ix->term_first_block, &ix->term_s0, &ix->blk_min_doc,
                                  &ix->blk_max_doc, &ix->blk_meta, &ix->blk_ub, &ix->blob, &ix->post_fn,
                                  &ix->doc_payload, &ix->s1})
        ix->device_bytes += b->bytes;
    *out = ix.release();
    return VBM25_OK;
}

void vbm25_index_destroy(vbm25_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->scratch) vbm25_batch_destroy(ix->scratch);
    delete ix;
}

uint64_t vbm25_index_device_bytes(const vbm25_index *ix) { return ix ? ix->device_bytes : 0; }

int vbm25_lookup_terms(const vbm25_index *ix, const uint8_t *keys, uint32_t n, uint32_t *term_ids) {
    if (!ix || (!keys && n) || (!term_ids && n)) return set_error(VBM25_ERR_INVALID, "NULL argument");
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t lo = 0, hi = ix->n_terms;
        const uint8_t *key = keys + 16ull * i;
        while (lo < hi) {
            uint32_t mid = (lo + hi) >> 1;
            if (std::memcmp(ix->term_key.data() + 16ull * mid, key, 16) < 0) lo = mid + 1; else hi = mid;
        }
        term_ids[i] = (lo < ix->n_terms && !std::memcmp(ix->term_key.data() + 16ull * lo, key, 16))
                          ? lo : UINT32_MAX;
    }
    return VBM25_OK;
}

static int vbm25_batch_create_impl(vbm25_index *ix, uint32_t max_queries, uint32_t max_total_terms, uint32_t k,
                       vbm25_batch **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!ix) return set_error(VBM25_ERR_INVALID, "index is NULL");
    if (k == 0) return set_error(VBM25_ERR_INVALID, "number of needed rows is set to 0");  // default.rs:114-116
    if (k > 65535) return set_error(VBM25_ERR_INVALID, "k exceeds bm25.limit's maximum of 65535");
    if (k > 1024)
        return set_error(VBM25_ERR_UNSUPPORTED, "k = %u: the GPU path currently keeps at most 1024 hits per query", k);
    if (!max_queries) return set_error(VBM25_ERR_INVALID, "max_queries is 0");
    if (int rc = use_device(ix->device)) return rc;
    auto bt = std::make_unique<vbm25_batch>();
    bt->index = ix;
    bt->max_queries = max_queries;
    bt->max_terms = max_total_terms;
    bt->k = k;
    {
        const char *env = std::getenv("VBM25_NO_CURSOR");
        bt->use_cursor = k <= (uint32_t)REG_K && !(env && env[0] == '1');
        const char *ti = std::getenv("VBM25_CUR_ITEMS");
        bt->target_items = bt->use_cursor ? (ti ? (uint32_t)std::atoi(ti) : CUR_TARGET_ITEMS) : TARGET_ITEMS;
        if (bt->target_items < TARGET_ITEMS) bt->target_items = TARGET_ITEMS;
        const char *mc = std::getenv("VBM25_CUR_MIN_CHUNK");
        bt->min_chunk = bt->use_cursor ? (mc ? (uint32_t)std::atoi(mc) : CUR_MIN_CHUNK_POSTINGS) : MIN_CHUNK_POSTINGS;
        if (bt->min_chunk < 128) bt->min_chunk = 128;
    }
    bt->max_items = max_queries + bt->target_items;
    int rc = 0;
    if ((rc = bt->term_ids.alloc(4ull * max_total_terms)) ||
        (rc = bt->q_off.alloc(4ull * (max_queries + 1))) ||
        (rc = bt->items.alloc(sizeof(Item) * size_t(bt->max_items))) || (rc = bt->n_items.alloc(4)) ||
        (rc = bt->q_item_base.alloc(4ull * (max_queries + 1))) ||
        (rc = bt->theta.alloc(8ull * max_queries)) ||
        (rc = bt->res_score.alloc(8ull * bt->max_items * k)) ||
        (rc = bt->res_doc.alloc(4ull * bt->max_items * k)) ||
        (rc = bt->res_cnt.alloc(4ull * bt->max_items)) ||
        (rc = bt->hits.alloc(sizeof(vbm25_hit) * size_t(max_queries) * k)) ||
        (rc = bt->n_hits.alloc(4ull * max_queries)) || (rc = bt->error_flag.alloc(4)) ||
        (rc = bt->q_dense.alloc(max_queries)) ||
        (rc = bt->spill.alloc(size_t(TARGET_ITEMS) * 3 * 2 * C_POSTINGS * 16)) ||
        (rc = bt->item_failed.alloc(4ull * bt->max_items)) || (rc = bt->work_ctr.alloc(4)) ||
        (rc = bt->hist.alloc(4ull * CUR_HB * max_queries)))
        return rc;
    HIP_TRY(hipMemset(bt->error_flag.p, 0, 4));
#ifdef VBM25_PROFILE
    if (int rc2 = bt->prof.alloc(8ull * 33 * CUR_GRID)) return rc2;
    HIP_TRY(hipMemset(bt->prof.p, 0, 8ull * 33 * CUR_GRID));
#endif
    *out = bt.release();
    return VBM25_OK;
}

void vbm25_batch_destroy(vbm25_batch *bt) {
    if (!bt) return;
    (void)hipSetDevice(bt->index->device);
    delete bt;
}

static int vbm25_batch_set_queries_impl(vbm25_batch *bt, const uint32_t *term_ids, const uint32_t *q_off,
                            uint32_t nq) {
    if (!bt || !q_off) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (nq > bt->max_queries) return set_error(VBM25_ERR_INVALID, "%u queries exceed the batch capacity %u", nq, bt->max_queries);
    if (q_off[0] != 0) return set_error(VBM25_ERR_INVALID, "q_off[0] must be 0");
    bool many = false, mid = false;
    uint32_t cur_mt = 1;
    // Routing: the chain kernel is built for sparse queries; a query with many postings per
    // document (Zipf head terms) or more than CHAIN_MAX_TERMS indexed terms takes the
    // dense-window kernel.  Tuning knob: VBM25_DENSE_X1000 (postings per 1000 documents).
    const char *env = std::getenv("VBM25_DENSE_X1000");
    const unsigned long long dense_x1000 = env ? (unsigned long long)std::atoll(env) : 100ull;
    std::vector<uint8_t> dense(nq, 0);
    for (uint32_t q = 0; q < nq; ++q) {
        if (q_off[q + 1] < q_off[q]) return set_error(VBM25_ERR_INVALID, "q_off not monotone at query %u", q);
        uint32_t valid = 0;
        unsigned long long postings = 0;
        for (uint32_t p = q_off[q]; p < q_off[q + 1]; ++p) {
            if (p > q_off[q] && term_ids[p] <= term_ids[p - 1])  // Query::checked_new, vector.rs:106-110
                return set_error(VBM25_ERR_INVALID, "query %u: term ids must be strictly ascending", q);
            valid += term_ids[p] < bt->index->n_terms;
            if (term_ids[p] < bt->index->n_terms) postings += bt->index->term_df_host[term_ids[p]];
        }
        if (postings * 1000ull >= dense_x1000 * bt->index->n_docs) {
            dense[q] = 1;
            many = true;
        }
        many |= valid > (uint32_t)CHAIN_MAX_TERMS;
        if (!dense[q]) {
            if (valid <= (uint32_t)CUR_T) cur_mt = std::max(cur_mt, valid);
            else if (valid <= (uint32_t)CHAIN_MAX_TERMS) mid = true;
        }
        if (valid > MAX_TERMS)
            return set_error(VBM25_ERR_UNSUPPORTED, "query %u has %u indexed terms; the GPU path handles up to %d", q, valid, MAX_TERMS);
    }
    if (q_off[nq] > bt->max_terms) return set_error(VBM25_ERR_INVALID, "%u terms exceed the batch capacity %u", q_off[nq], bt->max_terms);
    if (int rc = use_device(bt->index->device)) return rc;
    if (q_off[nq]) HIP_TRY(hipMemcpy(bt->term_ids.p, term_ids, 4ull * q_off[nq], hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(bt->q_off.p, q_off, 4ull * (nq + 1), hipMemcpyHostToDevice));
    if (nq) HIP_TRY(hipMemcpy(bt->q_dense.p, dense.data(), nq, hipMemcpyHostToDevice));
    bt->nq = nq;
    bt->has_many_terms = many;
    bt->has_mid_terms = mid;
    bt->cur_mt = cur_mt;
    return VBM25_OK;
}

static int vbm25_batch_run_impl(vbm25_batch *bt, void *hip_stream) {
    if (!bt) return set_error(VBM25_ERR_INVALID, "batch is NULL");
    if (!bt->nq) return VBM25_OK;
    if (int rc = use_device(bt->index->device)) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    DevBatch db{};
    db.term_ids = bt->term_ids.as<uint32_t>();
    db.q_off = bt->q_off.as<uint32_t>();
    db.nq = bt->nq;
    db.k = bt->k;
    db.items = bt->items.as<Item>();
    db.n_items = bt->n_items.as<uint32_t>();
    db.q_item_base = bt->q_item_base.as<uint32_t>();
    db.theta = bt->theta.as<unsigned long long>();
    db.res_score = bt->res_score.as<double>();
    db.res_doc = bt->res_doc.as<uint32_t>();
    db.res_cnt = bt->res_cnt.as<uint32_t>();
    db.hits = bt->hits.as<vbm25_hit>();
    db.n_hits = bt->n_hits.as<uint32_t>();
    db.error_flag = bt->error_flag.as<uint32_t>();
    db.q_dense = bt->q_dense.as<uint8_t>();
    db.spill = bt->spill.as<unsigned long long>();
    db.item_failed = bt->item_failed.as<uint32_t>();
    db.prof = bt->prof.as<unsigned long long>();
    db.hist = bt->hist.as<uint32_t>();
    db.work_ctr = bt->work_ctr.as<uint32_t>();
    db.chain_min_terms = bt->use_cursor ? (uint32_t)CUR_T + 1u : 0u;
    const DevIndex &ix = bt->index->dev;
    if (bt->use_cursor) HIP_TRY(hipMemsetAsync(bt->hist.p, 0, 4ull * CUR_HB * bt->nq, st));
    plan_kernel<<<1, PLAN_WG, 0, st>>>(ix, db, bt->max_items, bt->target_items, bt->min_chunk);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (bt->timing) {
        if (bt->events_used == bt->events.size()) {
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            bt->events.emplace_back(e0, e1);
        }
        e0 = bt->events[bt->events_used].first;
        e1 = bt->events[bt->events_used].second;
        bt->events_used++;
        HIP_TRY(hipEventRecord(e0, st));
    }
    const uint32_t grid = std::min<uint32_t>(bt->max_items, TARGET_ITEMS);
    const int rc = dispatch_k(bt->k, [&](auto kmax) {
        constexpr int KM = decltype(kmax)::value;
        if constexpr (KM <= REG_K) {
            if (bt->use_cursor) {
                // persistent single-wave workgroups; items are handed out through bt.work_ctr
                scan_cursor_kernel<KM><<<CUR_GRID, 64, 4 * cur_lds_words(bt->cur_mt), st>>>(ix, db, bt->cur_mt);
            }
        }
        if (!bt->use_cursor || bt->has_mid_terms) scan_kernel<KM><<<grid, CWG, 0, st>>>(ix, db);
        if (bt->timing) HIP_TRY(hipEventRecord(e1, st));
        // many-term / dense queries, and items the chain kernel gave up on (empty launch: 5 us)
        scan_many_kernel<decltype(kmax)::value><<<grid, WG, 0, st>>>(ix, db);
        merge_kernel<decltype(kmax)::value><<<bt->nq, 64, 0, st>>>(ix, db);
        return int(VBM25_OK);
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return VBM25_OK;
}

static int vbm25_batch_fetch_impl(vbm25_batch *bt, vbm25_hit *hits, uint32_t *n_hits) {
    if (!bt || (!hits && bt->nq) || (!n_hits && bt->nq)) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (int rc = use_device(bt->index->device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    uint32_t flag = 0;
    HIP_TRY(hipMemcpy(&flag, bt->error_flag.p, 4, hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(hipMemset(bt->error_flag.p, 0, 4));
        return set_error(VBM25_ERR_DEVICE, "device-side planner overflow (flag %u)", flag);
    }
    if (bt->nq) {
        HIP_TRY(hipMemcpy(hits, bt->hits.p, sizeof(vbm25_hit) * size_t(bt->nq) * bt->k, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(n_hits, bt->n_hits.p, 4ull * bt->nq, hipMemcpyDeviceToHost));
    }
    return VBM25_OK;
}

int vbm25_batch_device_results(vbm25_batch *bt, void **hits, void **n_hits) {
    if (!bt) return set_error(VBM25_ERR_INVALID, "batch is NULL");
    if (hits) *hits = bt->hits.p;
    if (n_hits) *n_hits = bt->n_hits.p;
    return VBM25_OK;
}

int vbm25_batch_set_timing(vbm25_batch *bt, int enabled) {
    if (!bt) return set_error(VBM25_ERR_INVALID, "batch is NULL");
    bt->timing = enabled != 0;
    bt->events_used = 0;
    return VBM25_OK;
}

int vbm25_batch_kernel_ms(vbm25_batch *bt, double *avg_ms, uint32_t *n_launches) {
    if (!bt || !avg_ms) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (int rc = use_device(bt->index->device)) return rc;
    double sum = 0;
    for (size_t i = 0; i < bt->events_used; ++i) {
        HIP_TRY(hipEventSynchronize(bt->events[i].second));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, bt->events[i].first, bt->events[i].second));
        sum += ms;
    }
    *avg_ms = bt->events_used ? sum / double(bt->events_used) : 0.0;
    if (n_launches) *n_launches = uint32_t(bt->events_used);
    bt->events_used = 0;
    return VBM25_OK;
}

#ifdef VBM25_PROFILE
int vbm25_scan_occupancy(void) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, scan_kernel<64>, CWG, 0) != hipSuccess) return -1;
    return blocks;
}
// profiling builds only (not declared in include/vbm25.h): copy out the phase counters
int vbm25_batch_profile(vbm25_batch *bt, unsigned long long *out, uint32_t n_workgroups) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, bt->prof.p, 8ull * 33 * n_workgroups, hipMemcpyDeviceToHost));
    return VBM25_OK;
}
#endif

static int vbm25_search_batch_impl(vbm25_index *ix, const uint32_t *term_ids, const uint32_t *q_off, uint32_t nq,
                       uint32_t k, vbm25_hit *hits, uint32_t *n_hits) {
    if (!ix || !q_off) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (nq == 0) return k ? VBM25_OK : set_error(VBM25_ERR_INVALID, "number of needed rows is set to 0");
    // the index keeps one batch object for this convenience entry point and re-uses it while
    // the shape fits (device buffers of a batch are far more expensive to create than a search)
    vbm25_batch *bt = ix->scratch;
    const uint32_t n_terms = q_off[nq] ? q_off[nq] : 1;
    if (!bt || bt->k != k || bt->max_queries < nq || bt->max_terms < n_terms) {
        if (bt) vbm25_batch_destroy(bt);
        ix->scratch = nullptr;
        if (int rc = vbm25_batch_create(ix, std::max(nq, 16u), std::max(n_terms, 256u), k, &bt)) return rc;
        ix->scratch = bt;
    }
    int rc = vbm25_batch_set_queries(bt, term_ids, q_off, nq);
    if (!rc) rc = vbm25_batch_run(bt, nullptr);
    if (!rc) rc = vbm25_batch_fetch(bt, hits, n_hits);
    return rc;
}

int vbm25_index_create(const vbm25_index_desc *d, int device, vbm25_index **out) {
    return guarded([&] { return vbm25_index_create_impl(d, device, out); });
}

int vbm25_batch_create(vbm25_index *ix, uint32_t max_queries, uint32_t max_total_terms, uint32_t k,
                       vbm25_batch **out) {
    return guarded([&] { return vbm25_batch_create_impl(ix, max_queries, max_total_terms, k, out); });
}

int vbm25_batch_set_queries(vbm25_batch *bt, const uint32_t *term_ids, const uint32_t *q_off,
                            uint32_t nq) {
    return guarded([&] { return vbm25_batch_set_queries_impl(bt, term_ids, q_off, nq); });
}

int vbm25_batch_run(vbm25_batch *bt, void *hip_stream) {
    return guarded([&] { return vbm25_batch_run_impl(bt, hip_stream); });
}

int vbm25_batch_fetch(vbm25_batch *bt, vbm25_hit *hits, uint32_t *n_hits) {
    return guarded([&] { return vbm25_batch_fetch_impl(bt, hits, n_hits); });
}

int vbm25_search_batch(vbm25_index *ix, const uint32_t *term_ids, const uint32_t *q_off, uint32_t nq,
                       uint32_t k, vbm25_hit *hits, uint32_t *n_hits) {
    return guarded([&] { return vbm25_search_batch_impl(ix, term_ids, q_off, nq, k, hits, n_hits); });
}

}  // extern "C"
