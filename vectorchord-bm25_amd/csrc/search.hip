// search.hip -- device half of libvbm25: batched BM25 top-k over compressed posting blocks for MI355X
// (gfx950, wave64).  Replaces the traversal of /root/reference/crates/bm25/src/search.rs:28-282
// (bm25::search) for the sealed segment.  One translation unit; the kernels live in headers:
//   plan.h         post_fn_kernel (index preparation: per-posting fieldnorm stream + validation of block
//                  structure and WAND bounds) and plan_kernel (queries -> doc-range work items)
//   scan_win.h     scan_win_kernel (its own translation unit, scan_win.hip): sparse queries of <= 8 terms of comparable length, k <= 256
//                  (k <= 64 beyond five terms) -- the document-window formulation, the dominant kernel of C3
//   scan_range.h   scan_range_kernel: every other sparse query of <= 16 terms, k <= 256; the one-launch route of vbm25_search_batch (C2)
//   scan_dense.h   scan_dense_kernel: queries with many postings per document (Zipf head terms; C5), <= 16 terms, k <= 256
//   scan_many.h    scan_many_kernel: up to 1024 terms, 256 < k <= 1024, items the others gave up (exhaustive)
//   merge.h        merge_kernel: per-item top-k lists -> hits with payloads
//   decode.h / block_fetch.h / topk_lds.h / topk_reg.h / device_types.h   shared pieces
// This file: error text, host objects (index, batch, the pipelined ring vbm25_stream, the multi-device handle with its host
// threads), routing, and the C ABI of include/vbm25.h.  DESIGN.md has the full story.
//
// Result order is canonical: score descending, ties by ascending doc id.  All f64 arithmetic is IEEE
// (compiled with -ffp-contract=off, no fast-math): results are bit-identical to the CPU oracle's
// brute-force evaluation.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "vbm25_internal.h"
#include "device_segment.h"

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

#include "device_types.h"
#include "decode.h"
#include "plan.h"
#include "topk_lds.h"
#include "block_fetch.h"
#include "topk_reg.h"
#include "decode_id16.h"
#include "scan_range.h"
#include "scan_win_launch.h"
#include "scan_dense.h"
#include "scan_many.h"
#include "merge.h"

// ---------------------------------------------------------------------------
// Batched `<&>`: bm25::evaluate (evaluate.rs:22-74) for many documents against one query -- the seq-scan
// form of the operator (src/index/operators.rs:22-55).  One thread per document: a merge of the document's
// elements with the query's terms, both ascending; result = sum of idf * tf in query key order.  idf comes
// from the host (libm log, bm25.rs:285-289), tf() is bm25.rs:291-295 with the index's s1 table (the same
// expression), the fieldnorm of the document is length_to_fieldnorm of its saturating sum of tfs.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// k > 1024 (bm25.limit goes up to 65535, gucs.rs:37-46): exhaustive path, one query at a time.  acc[d] is the
// score of document d: one launch per term in ascending key order adds that term's postings (a document has
// at most one posting per term, so the adds of a launch never collide and the sum order is the key order of
// evaluate.rs:43-72).  Positive doubles order like their bit patterns (crates/score/src/lib.rs:46-60), so a
// stable descending radix sort of (bits(acc[d]), d) over all documents gives score descending, ties by
// ascending id; the first k entries with a non-zero key are the result (Results, search.rs:284-314).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bigk_accum_kernel(DevIndex ix, uint32_t term, double *acc) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t j0 = ix.term_first_block[term], j1 = ix.term_first_block[term + 1];
    const double s0 = ix.term_s0[term];
    for (uint32_t j = j0 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); j < j1; j += gridDim.x * (blockDim.x / 64)) {
        const uint4 bm = ix.blk_meta[j];
        const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
        const uint8_t *body = ix.blob + 8ull * bm.z;
        uint32_t d0, d1, f0, f1;
        decode_doc_ids(body, md, n, bm.x, lane, d0, d1);
        decode_fields(body + ((payload_bytes(md, n) + 7u) & ~7u), mt, n, lane, f0, f1);
        const uchar2 fn = reinterpret_cast<const uchar2 *>(ix.post_fn + 128ull * j)[lane];
        if (2 * lane < n) {
            const double tf = (double)f0;
            acc[d0] = acc[d0] + (tf * s0) / (tf + ix.s1[fn.x]);  // Cache::evaluate, bm25.rs:355-358
        }
        if (2 * lane + 1 < n) {
            const double tf = (double)f1;
            acc[d1] = acc[d1] + (tf * s0) / (tf + ix.s1[fn.y]);
        }
    }
}
__global__ void __launch_bounds__(256) bigk_iota_kernel(uint32_t *v, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}
__global__ void __launch_bounds__(256) bigk_emit_kernel(DevIndex ix, const unsigned long long *keys, const uint32_t *docs,
                                                        uint32_t n_docs, uint32_t k, vbm25_hit *hits, uint32_t *n_hits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && (n_docs == 0 || keys[0] == 0)) *n_hits = 0;
    if (i >= k || i >= n_docs) return;
    const unsigned long long key = keys[i];
    if (key == 0) return;
    const uint32_t d = docs[i];
    const uint16_t *pl = ix.doc_payload + 3ull * d;
    unsigned long long *out = reinterpret_cast<unsigned long long *>(hits + i);
    out[0] = key;
    out[1] = (unsigned long long)d | (unsigned long long)pl[0] << 32 | (unsigned long long)pl[1] << 48;
    out[2] = (unsigned long long)pl[2];
    if (i + 1 == k || i + 1 == n_docs || keys[i + 1] == 0) *n_hits = i + 1;
}

struct EvalArgs {
    uint32_t n_docs, n_q, n_terms;
    const uint32_t *q_terms;     // ascending term ids; ids >= n_terms (unknown tokens) are skipped
    const uint64_t *doc_start;   // n_docs + 1
    const uint32_t *doc_term;    // per element: term id, NONE32 when the key is not in the index
    const uint32_t *doc_tf;
    const double *term_idf, *s1;
    const uint32_t *fn_len;      // FIELDNORM_TO_LENGTH, 256 entries
    double k1p1;
    double *out;
};
__global__ void __launch_bounds__(256) evaluate_kernel(EvalArgs a) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= a.n_docs) return;
    const uint64_t e0 = a.doc_start[d], e1 = a.doc_start[d + 1];
    unsigned long long length = 0;  // Document::length, vector.rs:77-83: saturating
    for (uint64_t e = e0; e < e1; ++e) {
        length += a.doc_tf[e];
        if (length > 0xffffffffull) length = 0xffffffffull;
    }
    uint32_t lo = 0, hi = 256;  // length_to_fieldnorm, bm25.rs:278-283: last entry <= length
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.fn_len[mid] <= (uint32_t)length) lo = mid; else hi = mid;
    }
    const double s1 = a.s1[lo];
    uint64_t cur = e0;
    double result = 0.0;
    for (uint32_t i = 0; i < a.n_q; ++i) {
        const uint32_t qt = a.q_terms[i];
        if (qt >= a.n_terms) continue;
        while (cur < e1 && (a.doc_term[cur] >= a.n_terms || a.doc_term[cur] < qt)) ++cur;
        if (!(cur < e1 && a.doc_term[cur] == qt)) continue;
        const double tf = (double)a.doc_tf[cur];
        const double tfv = (tf * a.k1p1) / (tf + s1);
        result += a.term_idf[qt] * tfv;
    }
    a.out[d] = result;
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
    uint32_t n_docs = 0, n_terms = 0, n_blocks = 0;  // n_docs == 0: empty sealed segment, every search returns no hit
    std::vector<uint8_t> term_key;  // host copy for vbm25_lookup_terms
    std::vector<uint32_t> term_df_host;  // host copy for query routing
    vbm25_batch *scratch = nullptr;      // batch object re-used by vbm25_search_batch
    DeviceBuffer term_wand_tf, term_wand_fn, term_df, term_first_block, term_s0, blk_min_doc, blk_max_doc, blk_meta, blk_ub, blob,
        post_fn, post_rel16, post_tfn, doc_payload, s1, term_idf, fn_len, term_kth_ub, post_id16, win_off, term_win;
    std::vector<uint32_t> term_win_host;  // host copy of term_win (query routing); empty: the index has no window planes
    uint32_t n_win = 0;
    double k1 = 1.2;
    uint64_t device_bytes = 0;
};

// Tuning / test switches (not part of the ABI of include/vbm25.h; set through vbm25_tuning_set by tools and tests, read when a
// batch object is created).  No entry point of the library reads the environment.
struct Tuning {
    long long dense_x1000 = 100;   // a query with this many postings per 1000 documents is dense (0: every query)
    int dense = 1;                 // dense queries take scan_dense_kernel (0: the exhaustive scan_many_kernel)
    int ne = 1;                    // MaxScore split (non-essential lists looked up, not scanned)
    int fused = 1;                 // one-launch route for a handful of sparse queries
    uint32_t ne_ratio = 2;
    uint32_t dense_items = D_TARGET_ITEMS;
    uint32_t range_items = R_TARGET_ITEMS, range_min_chunk = R_MIN_CHUNK_POSTINGS;
    uint32_t range_grid = R_GRID, dense_grid = D_GRID;
    uint32_t dbg = 0;              // timing experiments of scan_win_kernel only (wrong results)
    uint32_t fused_items = 128;    // the one-launch route takes batches of up to this many work items
    int arith = 1;                 // batches of sparse queries beyond that: work items made by the scan kernel (no plan_kernel)
    int win = 1;                   // batches of sparse queries of <= 8 comparable terms, k <= 64: scan_win_kernel (the window formulation)
    int win_force = 0;             // tests: that route whatever the lists' lengths
    uint32_t win_items = 0;        // work items of a batch on that route (0: twice the resident waves)
    int win_planes = 1;            // read at index creation: derive the window planes (post_id16, win_off)
    int rel16_plane = 1;           // read at index creation: derive post_rel16 (0: the kernels decode the blob's delta streams themselves)
    int id16_plane = 1;            // read at index creation: derive post_id16 (0: the window tables only -- decode_id16_kernel unpacks the batch's
                                   // terms from the blob into the batch's scratch plane ahead of every scan_win_kernel launch)
    int win_guided = 0;            // scan_win_kernel's items of a query of decreasing length, handed out longest first (0: equal runs; measured
                                   // no better on C3 -- an item's setup costs more than the shorter tail saves)
    uint32_t win_grid = 0;         // its persistent workgroups (0: one per CU)
    int win_fuse = 1;              // scan_win_kernel merges the queries' lists itself: the batched route is one launch (0: scan_many_kernel and merge_kernel behind it)
    int win_cut1 = 392, win_cut2 = 730;  // ... where a query's three runs are cut, in thousandths of its windows (round 6, tools/skew_sweep.sh: 59 / 52 / 42 of C3's 153)
    int win_order_arith = 1;       // the skewed layout computed in the kernel when the queries keep their order (0: always the host's table)
    int win_skew = 1;              // one item per wave: a query's three runs of windows sized for the three kinds of waves of a SIMD
    uint32_t generation = 0;       // bumped by every vbm25_tuning_set / reset: vbm25_search_batch's batch object is rebuilt when it is stale
};
static Tuning g_tune;
static std::mutex g_tune_mutex;  // (set / reset / the copy a new batch takes)
static Tuning tuning_snapshot() {
    std::lock_guard<std::mutex> g(g_tune_mutex);
    return g_tune;
}

struct vbm25_batch {
    vbm25_index *index = nullptr;
    int device = 0;  // the index's device ordinal: the batch can be destroyed after its index
    uint32_t max_queries = 0, max_terms = 0, k = 0, nq = 0, max_items = 0;
    DeviceBuffer term_ids, q_off, items, n_items, q_item_base, theta, res_score, res_doc, res_cnt,
        hits, n_hits, error_flag, prof, q_dense, item_failed, item_order, work_ctr, hist, fused_state, dbg, fail_any, q_failed, theta_last;
    bool bigk = false;            // k > 1024: exhaustive path, one query at a time
    DeviceBuffer bk_acc, bk_keys, bk_iota, bk_docs, bk_tmp;
    size_t bk_tmp_bytes = 0;
    std::vector<uint32_t> h_terms, h_off;  // host copy of the queries (bigk launches per term)
    std::vector<uint8_t> h_dense;          // per query: dense (scratch of set_queries, sized once)
    std::vector<unsigned long long> h_postings;
    std::vector<uint32_t> h_order, h_order_q;  // set_queries: the longest-first item order of the route without plan_kernel
    Tuning tune;                  // the switches of the moment the batch was created
    bool timing = false;
    bool use_range = false;       // k <= REG_K: sparse queries of <= 16 terms take scan_range_kernel, dense ones scan_dense_kernel
    uint32_t range_rt = 0;        // ... with this row stride (8 or 16) for the current queries; 0 = not used
    uint32_t lpi = 1;             // result lists per work item
    uint32_t range_grid = R_GRID;
    bool use_dense = false;       // dense queries of <= D_T terms take scan_dense_kernel
    bool has_dense = false;       // ... and the current queries have such a query
    uint32_t dense_grid = D_GRID;
    uint32_t dense_c = 0;         // items per dense query of the current queries (0: chunks by postings, as the other queries)
    // vbm25_search_batch with a handful of sparse queries: ONE launch (scan_range_kernel plans, scans and merges)
    uint32_t fused_g = 0;         // items per query of the current queries on that route (0: general route)
    uint32_t arith_g = 0;         // general route without plan_kernel (every query sparse): items per query, made by the scan kernel itself
    uint32_t win_mt = 8;          // scan_win_kernel: the most indexed terms of a query of the current batch
    bool win_skew = false;        // scan_win_kernel: one item per wave, a query's three runs sized for the three kinds of waves of a SIMD
    uint32_t q_stride = 0;        // != 0: every query of the current batch has this many terms
    bool order_identity = false;  // ... and their order in the host's item order is the caller's (no sort by length)
    bool order_useful = true;     // the current queries differ enough in length for the longest-first order to matter
    uint32_t win_g = 0;           // ... and scan_win_kernel's flavour of it (items = runs of 2^16-document windows, one result list each)
    uint32_t win_len = 0;         // ... a query's runs: win_len windows each and a shorter rest (0: equal runs)
    // an index without the post_id16 plane: the batch's scratch plane (the low 16 bits of the ids of the current queries' terms, made by
    // decode_id16_kernel ahead of every scan_win_kernel launch) and, per term position, the term's first 256-byte block in it
    DeviceBuffer id16_tmp, id16_fb;
    std::vector<uint32_t> h_id16_fb;
    bool id16_decode = false;     // the current queries take scan_win_kernel through the scratch plane
    uint32_t n_term_pos = 0;      // term positions of the current queries (q_off[nq])
    size_t pin_id16_bytes = 0;    // ... and their h_id16_fb is staged behind the item order (upload_staged)
    DeviceBuffer qin;             // the staged descriptors on the device, one block: term ids | offsets | dense flags (padded to 8) | item order
    bool qin_live = false;        // ... hold the current queries (set by upload_staged; a plain set_queries fills the separate buffers)
    bool device_consumer = false; // vbm25_batch_device_results was called: every run leaves complete records on the device
    bool win_nofuse = false;      // the last run's in-kernel merge marked a query (an item was given up): this query set runs with scan_many_kernel and merge_kernel
    bool win_fused_run = false;   // the last run was a one-launch run of scan_win_kernel (a count of NONE32 means: re-run, see vbm25_batch_fetch_impl)
    bool need_many = true;        // the current queries have items for scan_many_kernel (more than 16 terms, 256 < k, dense without the dense kernel)
    bool fused_pinned = false;    // ... with queries and hits in pinned host memory (vbm25_search_batch, <= 8 queries); else device buffers
    bool state_clean = false;     // threshold / histogram / counters are zero (the fused route leaves them so; the general one does not)
    uint32_t target_items = TARGET_ITEMS;
    uint32_t min_chunk = MIN_CHUNK_POSTINGS;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    // vbm25_search_batch's low-latency route: pinned staging buffers and a private stream -- queries go up and
    // hits come down with asynchronous copies and ONE stream synchronisation
    hipStream_t lat_stream = nullptr;
    hipStream_t last_stream = nullptr;  // the stream of the last run: what fetch waits for (not the whole device)
    bool download_enqueued = false;     // the last run's records are already on their way to pin_out (vbm25_multi_batch_run)
    // vbm25_stream_*: merge_kernel writes counts and records straight into the pinned output buffer (posted writes over PCIe, no
    // download command on the step: only the 4-byte flag is copied); results_pinned_now: the last run did so
    bool pinned_results = false, results_pinned_now = false;
    uint8_t *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_bytes = 0, pin_out_bytes = 0, pin_nt = 0, pin_order_bytes = 0;
    ~vbm25_batch() {
        if (lat_stream) (void)hipStreamDestroy(lat_stream);
        if (pin_in) (void)hipHostFree(pin_in);
        if (pin_out) (void)hipHostFree(pin_out);
        for (auto &e : events) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
    }
};


namespace {
// every device array of an index (vbm25_multi_create copies them from GPU to GPU)
DeviceBuffer vbm25_index::*const INDEX_BUFFERS[] = {
    &vbm25_index::term_wand_tf, &vbm25_index::term_wand_fn, &vbm25_index::term_df, &vbm25_index::term_first_block, &vbm25_index::term_s0,
    &vbm25_index::blk_min_doc, &vbm25_index::blk_max_doc, &vbm25_index::blk_meta, &vbm25_index::blk_ub, &vbm25_index::blob,
    &vbm25_index::post_fn, &vbm25_index::post_rel16, &vbm25_index::post_tfn, &vbm25_index::doc_payload, &vbm25_index::s1,
    &vbm25_index::term_idf, &vbm25_index::fn_len, &vbm25_index::term_kth_ub,
    &vbm25_index::post_id16, &vbm25_index::win_off, &vbm25_index::term_win};

void fill_dev(vbm25_index *ix) {
    ix->dev.n_docs = ix->n_docs;
    ix->dev.n_terms = ix->n_terms;
    ix->dev.n_blocks = ix->n_blocks;
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
    ix->dev.post_rel16 = ix->post_rel16.as<uint32_t>();
    ix->dev.post_tfn = ix->post_tfn.as<uint32_t>();
    ix->dev.doc_payload = ix->doc_payload.as<uint16_t>();
    ix->dev.s1 = ix->s1.as<double>();
    ix->dev.term_kth_ub = ix->term_kth_ub.as<double>();  // (NULL when the block maxima are not attained)
    ix->dev.post_id16 = ix->post_id16.as<uint32_t>();    // (the three of them NULL when the index has no window planes)
    ix->dev.win_off = ix->win_off.as<uint32_t>();
    ix->dev.term_win = ix->term_win.as<uint32_t>();
    ix->dev.n_win = ix->n_win;
}
}  // namespace

namespace {

constexpr int VBM25_RETRY_GENERAL = 1000;  // internal: never leaves this file

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

// A flattened sealed segment as index creation sees it: the big arrays on the host (vbm25_index_desc) or in the HBM of the
// index's device (vbm25_device_segment); the vocabulary-sized ones the host computes with (libm log) always on the host.
struct RawSegment {
    bool on_device;
    uint32_t n_docs, n_terms, n_blocks;
    uint64_t sum_len, blob_bytes;
    double k1, b;
    const uint8_t *term_key;                                  // host
    const uint32_t *term_df_host, *term_first_block_host;     // host
    const uint32_t *term_df, *term_wand_tf, *term_first_block, *blk_min_doc, *blk_max_doc, *blk_wand_tf, *blk_off8;
    const uint8_t *term_wand_fn, *blk_n, *blk_wand_fn, *blk_meta_doc, *blk_meta_tf, *blob, *doc_fieldnorm;
    const uint16_t *doc_payload;
};

static int index_create_common(const RawSegment &r, int device, vbm25_index **out) {
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
    ix->n_docs = r.n_docs;
    ix->n_terms = r.n_terms;
    ix->n_blocks = r.n_blocks;
    ix->k1 = r.k1;
    ix->term_key.assign(r.term_key, r.term_key + 16ull * r.n_terms);
    ix->term_df_host.assign(r.term_df_host, r.term_df_host + r.n_terms);
    const hipMemcpyKind kind = r.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    auto put = [&](DeviceBuffer &dst, const void *src, size_t bytes) -> int {
        if (int rc = dst.alloc(bytes)) return rc;
        if (bytes) HIP_TRY(hipMemcpy(dst.p, src, bytes, kind));
        return VBM25_OK;
    };
    const bool has_wand = r.blk_wand_fn && r.blk_wand_tf;

    // per-term s0 = idf (k1 + 1), idf (vbm25_evaluate_batch) -- host libm log, bm25.rs:285-289,348 -- and the s1 table of
    // bm25.rs:349-352
    std::vector<double> s0(r.n_terms), idf(r.n_terms);
    for (uint32_t t = 0; t < r.n_terms; ++t) {
        s0[t] = bm25_s0(r.n_docs, r.term_df_host[t], r.k1);
        idf[t] = std::log((double(r.n_docs) + 1.0) / (double(r.term_df_host[t]) + 0.5));
    }
    double s1[256];
    bm25_tables(r.n_docs, r.sum_len, r.k1, r.b, s1);
    // scan_win_kernel's planes: the low 16 bits of every id in posting order, and for every term with at least a posting per four
    // windows the table of its n_win + 1 window offsets (a rarer term's table would be larger than its list)
    const Tuning tune_now = tuning_snapshot();
    const bool win_planes = tune_now.win_planes != 0 && r.n_blocks != 0;
    // (`rel16_plane` = 0: no post_rel16 -- scan_range_kernel and scan_dense_kernel unpack the delta streams of the blob in the
    // kernel: 256 bytes per block less in HBM, more instructions per block; DESIGN.md section 1 has both measured)
    const bool rel16_plane = tune_now.rel16_plane != 0;
    const bool id16_plane = tune_now.id16_plane != 0;  // (0: the tables without the plane -- decode_id16.h)
    const uint32_t n_win = uint32_t((uint64_t(r.n_docs) + 65535u) >> 16);
    std::vector<uint32_t> term_win(win_planes ? r.n_terms : 0u, UINT32_MAX);
    uint64_t n_woff = n_win + 2u;  // (entries 0 .. n_win + 1: the NULL table -- a term without postings, what a query's missing terms read)
    if (win_planes) {
        // BUDGET (round-5 advisor): a table is n_win + 1 words whatever the term's length -- up to four words per posting at the
        // threshold -- so on a corpus of thousands of windows and hundreds of thousands of qualifying terms the tables would take
        // gigabytes nobody asked for.  They get at most a quarter of what post_id16 takes (64 bytes per block; never less than 4 MiB), the
        // longest lists first: the routing sends a query through the window kernel only when EVERY term has a table, and the terms it
        // wants there are the long ones (32 .. 232 postings per window).
        const uint64_t budget_words = std::max<uint64_t>(1ull << 20, 16ull * r.n_blocks);
        std::vector<uint32_t> cand;
        for (uint32_t t = 0; t < r.n_terms; ++t)
            if (uint64_t(r.term_df_host[t]) * 4u >= n_win) cand.push_back(t);
        if (uint64_t(cand.size()) * (n_win + 1u) > budget_words) {
            std::stable_sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return r.term_df_host[a] > r.term_df_host[b]; });
            cand.resize(size_t(budget_words / (n_win + 1u)));
            std::sort(cand.begin(), cand.end());  // (tables in term order, as without a budget)
        }
        for (uint32_t t : cand) {
            if (n_woff + n_win + 1u > 0xfffffff0ull) break;
            term_win[t] = uint32_t(n_woff);
            n_woff += n_win + 1u;
        }
    }
    // the raw per-block arrays the derivation reads: used where they are (device segment) or uploaded for its duration
    DeviceBuffer t_n, t_wfn, t_wtf, t_md, t_mt, t_off8, t_fieldnorm, t_raw, t_sorted, t_tmp, err;
    const uint8_t *p_n = r.blk_n, *p_wfn = r.blk_wand_fn, *p_md = r.blk_meta_doc, *p_mt = r.blk_meta_tf, *p_fieldnorm = r.doc_fieldnorm;
    const uint32_t *p_wtf = r.blk_wand_tf, *p_off8 = r.blk_off8;
    int rc = 0;
    if (!r.on_device) {
        if ((rc = t_n.upload(r.blk_n, r.n_blocks)) || (rc = t_md.upload(r.blk_meta_doc, r.n_blocks)) ||
            (rc = t_mt.upload(r.blk_meta_tf, r.n_blocks)) || (rc = t_off8.upload(r.blk_off8, 4ull * (r.n_blocks + 1ull))) ||
            (rc = t_fieldnorm.upload(r.doc_fieldnorm, r.n_docs)) ||
            (has_wand && ((rc = t_wfn.upload(r.blk_wand_fn, r.n_blocks)) || (rc = t_wtf.upload(r.blk_wand_tf, 4ull * r.n_blocks)))))
            return rc;
        p_n = t_n.as<uint8_t>();
        p_md = t_md.as<uint8_t>();
        p_mt = t_mt.as<uint8_t>();
        p_off8 = t_off8.as<uint32_t>();
        p_fieldnorm = t_fieldnorm.as<uint8_t>();
        p_wfn = t_wfn.as<uint8_t>();
        p_wtf = t_wtf.as<uint32_t>();
    }
    // slack: the scan kernels read whole 256-byte LDS-DMA slots / word pairs from a block's first byte
    const size_t blob_alloc = ((size_t(r.blob_bytes) + 15) & ~size_t(15)) + 512;
    if ((rc = put(ix->term_df, r.term_df, 4ull * r.n_terms)) ||
        (rc = put(ix->term_first_block, r.term_first_block, 4ull * (r.n_terms + 1ull))) ||
        (rc = ix->term_s0.upload(s0.data(), 8ull * r.n_terms)) ||
        (rc = ix->term_idf.upload(idf.data(), 8ull * r.n_terms)) ||
        (rc = ix->fn_len.upload(fieldnorm_lengths(), 4ull * 256)) ||
        (rc = put(ix->term_wand_tf, r.term_wand_tf, 4ull * r.n_terms)) ||
        (rc = put(ix->term_wand_fn, r.term_wand_fn, r.n_terms)) ||
        (rc = put(ix->blk_min_doc, r.blk_min_doc, 4ull * r.n_blocks)) ||
        (rc = put(ix->blk_max_doc, r.blk_max_doc, 4ull * r.n_blocks)) ||
        (rc = ix->blk_meta.alloc(16ull * r.n_blocks)) ||
        (rc = ix->blk_ub.alloc(8ull * r.n_blocks)) ||
        (rc = ix->blob.alloc(blob_alloc)) ||
        (rc = ix->post_fn.alloc(128ull * r.n_blocks)) ||
        (rel16_plane && (rc = ix->post_rel16.alloc(256ull * r.n_blocks))) ||
        (rc = ix->post_tfn.alloc(256ull * r.n_blocks + 1024)) ||  // (slack: scan_win_kernel's cold pass reads whole runs)
        (win_planes && ((id16_plane && (rc = ix->post_id16.alloc(256ull * r.n_blocks + 1024))) || (rc = ix->win_off.alloc(4ull * n_woff)) ||
                        (rc = ix->term_win.upload(term_win.data(), 4ull * r.n_terms)))) ||
        (rc = put(ix->doc_payload, r.doc_payload, 6ull * r.n_docs)) ||
        (rc = ix->s1.upload(s1, sizeof s1)) || (rc = err.alloc(4)))
        return rc;
    HIP_TRY(hipMemset(err.p, 0, 4));
    HIP_TRY(hipMemset(ix->blob.p, 0, blob_alloc));
    if (r.blob_bytes) HIP_TRY(hipMemcpy(ix->blob.p, r.blob, r.blob_bytes, kind));
    if (r.n_blocks) {
        if (has_wand && ((rc = t_raw.alloc(8ull * r.n_blocks)) || (rc = t_sorted.alloc(8ull * r.n_blocks)) ||
                         (rc = ix->term_kth_ub.alloc(8ull * KTH_LEVELS * r.n_terms))))
            return rc;
        DeriveArgs da{};
        da.n_blocks = r.n_blocks;
        da.n_terms = r.n_terms;
        da.has_wand = has_wand ? 1u : 0u;
        da.term_first_block = ix->term_first_block.as<uint32_t>();
        da.term_wand_tf = ix->term_wand_tf.as<uint32_t>();
        da.term_wand_fn = ix->term_wand_fn.as<uint8_t>();
        da.blk_min_doc = ix->blk_min_doc.as<uint32_t>();
        da.blk_max_doc = ix->blk_max_doc.as<uint32_t>();
        da.blk_off8 = p_off8;
        da.blk_wand_tf = p_wtf;
        da.blk_n = p_n;
        da.blk_meta_doc = p_md;
        da.blk_meta_tf = p_mt;
        da.blk_wand_fn = p_wfn;
        da.term_s0 = ix->term_s0.as<double>();
        da.s1 = ix->s1.as<double>();
        da.blk_meta = ix->blk_meta.as<uint4>();
        da.blk_ub = ix->blk_ub.as<double>();
        da.blk_raw = t_raw.as<double>();
        blk_derive_kernel<<<(r.n_blocks + 255) / 256, 256>>>(da);
        HIP_TRY(hipGetLastError());
        if (has_wand) {
            // per term the 2^i-th largest block maximum, i = 0..8 (the scan kernels' first threshold: with block WAND pairs
            // every block maximum is the score of a posting of its block -- post_fn_kernel verifies that -- so k distinct
            // documents of the term score at least the k-th largest of them): a segmented sort of the maxima by term
            size_t tb = 0;
            const uint32_t *seg = ix->term_first_block.as<uint32_t>();
            HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(nullptr, tb, t_raw.as<double>(), t_sorted.as<double>(), (int)r.n_blocks,
                                                                        (int)r.n_terms, seg, seg + 1));
            if ((rc = t_tmp.alloc(tb))) return rc;
            HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeysDescending(t_tmp.p, tb, t_raw.as<double>(), t_sorted.as<double>(), (int)r.n_blocks,
                                                                        (int)r.n_terms, seg, seg + 1));
            kth_pick_kernel<<<(r.n_terms * KTH_LEVELS + 255) / 256, 256>>>(r.n_terms, seg, t_sorted.as<double>(), ix->term_kth_ub.as<double>());
            HIP_TRY(hipGetLastError());
        }
        const uint32_t grid = (r.n_blocks + 3) / 4;
        PostFnArgs pa{};
        pa.n_blocks = r.n_blocks;
        pa.n_docs = r.n_docs;
        pa.n_terms = r.n_terms;
        pa.blk_meta = ix->blk_meta.as<uint4>();
        pa.blob = ix->blob.as<uint8_t>();
        pa.doc_fieldnorm = p_fieldnorm;
        pa.post_fn = ix->post_fn.as<uint8_t>();
        pa.post_rel16 = ix->post_rel16.as<uint32_t>();
        pa.post_tfn = ix->post_tfn.as<uint32_t>();
        pa.post_id16 = ix->post_id16.as<uint32_t>();
        pa.win_off = ix->win_off.as<uint32_t>();
        pa.term_win = ix->term_win.as<uint32_t>();
        pa.n_win = n_win;
        if (win_planes) HIP_TRY(hipMemset(ix->win_off.p, 0, 4ull * n_woff));
        pa.error_flag = err.as<uint32_t>();
        pa.term_first_block = ix->term_first_block.as<uint32_t>();
        pa.term_wand_tf = ix->term_wand_tf.as<uint32_t>();
        pa.term_wand_fn = ix->term_wand_fn.as<uint8_t>();
        pa.term_s0 = ix->term_s0.as<double>();
        pa.s1 = ix->s1.as<double>();
        pa.blk_ub = ix->blk_ub.as<double>();
        pa.blk_raw = has_wand ? t_raw.as<double>() : nullptr;
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
    if (win_planes && !(flag & 8u)) {  // (flag 8: a short block inside a term -- postings are not at 128 block + i: no window planes)
        ix->term_win_host = std::move(term_win);
        ix->n_win = n_win;
    } else {
        for (DeviceBuffer *b : {&ix->post_id16, &ix->win_off, &ix->term_win}) {
            if (b->p) (void)hipFree(b->p);
            b->p = nullptr;
            b->bytes = 0;
        }
    }
    ix->dev.blk_ub_attained = has_wand && !(flag & 4u) ? 1u : 0u;
    if (!ix->dev.blk_ub_attained && ix->term_kth_ub.p) {  // the k-th largest maxima bound nothing then: kept out of the index (and its replicas)
        (void)hipFree(ix->term_kth_ub.p);
        ix->term_kth_ub.p = nullptr;
        ix->term_kth_ub.bytes = 0;
    }
    fill_dev(ix.get());
    ix->dev.blob_bytes = r.blob_bytes;
    ix->dev.blk_ub_attained = has_wand && !(flag & 4u) ? 1u : 0u;
    for (const DeviceBuffer *b : {&ix->term_df, &ix->term_first_block, &ix->term_s0, &ix->blk_min_doc,
                                  &ix->blk_max_doc, &ix->blk_meta, &ix->blk_ub, &ix->blob, &ix->post_fn,
                                  &ix->post_rel16, &ix->post_tfn, &ix->term_kth_ub, &ix->doc_payload, &ix->s1,
                                  &ix->post_id16, &ix->win_off, &ix->term_win})
        ix->device_bytes += b->bytes;
    *out = ix.release();
    return VBM25_OK;
}

static int vbm25_index_create_impl(const vbm25_index_desc *d, int device, vbm25_index **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (int rc = check_desc(d)) return rc;
    RawSegment r{};
    r.on_device = false;
    r.n_docs = d->n_docs;
    r.n_terms = d->n_terms;
    r.n_blocks = d->n_blocks;
    r.sum_len = d->sum_len;
    r.blob_bytes = d->blob_bytes;
    r.k1 = d->k1;
    r.b = d->b;
    r.term_key = d->term_key;
    r.term_df_host = r.term_df = d->term_df;
    r.term_first_block_host = r.term_first_block = d->term_first_block;
    r.term_wand_tf = d->term_wand_tf;
    r.term_wand_fn = d->term_wand_fn;
    r.blk_min_doc = d->blk_min_doc;
    r.blk_max_doc = d->blk_max_doc;
    r.blk_n = d->blk_n;
    r.blk_wand_fn = d->blk_wand_fn;
    r.blk_wand_tf = d->blk_wand_tf;
    r.blk_meta_doc = d->blk_meta_doc;
    r.blk_meta_tf = d->blk_meta_tf;
    r.blk_off8 = d->blk_off8;
    r.blob = d->blob;
    r.doc_fieldnorm = d->doc_fieldnorm;
    r.doc_payload = d->doc_payload;
    return index_create_common(r, device, out);
}

// The index of a segment that is already in HBM (vbm25_device_segment_synth / _build): device-to-device copies of the arrays the
// index keeps as they are, everything else derived where it lies.  The segment is left as it was.
static int vbm25_index_create_from_device_impl(const vbm25_device_segment *ds, vbm25_index **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!ds) return set_error(VBM25_ERR_INVALID, "segment is NULL");
    RawSegment r{};
    r.on_device = true;
    r.n_docs = ds->n_docs;
    r.n_terms = ds->n_terms;
    r.n_blocks = ds->n_blocks;
    r.sum_len = ds->sum_len;
    r.blob_bytes = ds->blob_bytes;
    r.k1 = ds->k1;
    r.b = ds->b;
    r.term_key = ds->term_key.data();
    r.term_df_host = ds->term_df.data();
    r.term_first_block_host = ds->term_first_block.data();
    r.term_df = ds->d_term_df.as<uint32_t>();
    r.term_first_block = ds->d_term_first_block.as<uint32_t>();
    r.term_wand_tf = ds->d_term_wand_tf.as<uint32_t>();
    r.term_wand_fn = ds->d_term_wand_fn.as<uint8_t>();
    r.blk_min_doc = ds->d_blk_min.as<uint32_t>();
    r.blk_max_doc = ds->d_blk_max.as<uint32_t>();
    r.blk_n = ds->d_blk_n.as<uint8_t>();
    r.blk_wand_fn = ds->d_blk_wand_fn.as<uint8_t>();
    r.blk_wand_tf = ds->d_blk_wand_tf.as<uint32_t>();
    r.blk_meta_doc = ds->d_blk_meta_doc.as<uint8_t>();
    r.blk_meta_tf = ds->d_blk_meta_tf.as<uint8_t>();
    r.blk_off8 = ds->d_blk_off8.as<uint32_t>();
    r.blob = ds->d_blob.as<uint8_t>();
    r.doc_fieldnorm = ds->d_doc_fieldnorm.as<uint8_t>();
    r.doc_payload = ds->d_doc_payload.as<uint16_t>();
    return index_create_common(r, ds->device, out);
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
    if (!max_queries) return set_error(VBM25_ERR_INVALID, "max_queries is 0");
    if (int rc = use_device(ix->device)) return rc;
    auto bt = std::make_unique<vbm25_batch>();
    bt->index = ix;
    bt->device = ix->device;
    bt->max_queries = max_queries;
    bt->max_terms = max_total_terms;
    bt->k = k;
    bt->tune = tuning_snapshot();
    bt->h_dense.resize(max_queries);
    bt->h_postings.resize(max_queries);
    // Routing by k: k <= 256 -- sparse queries of <= 16 terms: scan_range_kernel, dense ones: scan_dense_kernel, the rest and
    // whatever those two give up: scan_many_kernel; 256 < k <= 1024: scan_many_kernel (LDS top-k); above: the exhaustive path
    bt->use_range = k <= (uint32_t)REG_K;
    bt->lpi = bt->use_range ? (uint32_t)RNW : 1u;
    // (scan_dense_kernel reads post_rel16 unconditionally: an index made without the plane sends its dense queries to scan_many_kernel)
    bt->use_dense = bt->use_range && k <= (uint32_t)D_KMAX && bt->tune.dense != 0 && ix->post_rel16.p != nullptr;
    bt->target_items = bt->use_range ? std::max(256u, bt->tune.range_items) : TARGET_ITEMS;
    bt->min_chunk = bt->use_range ? std::max(128u, bt->tune.range_min_chunk) : MIN_CHUNK_POSTINGS;
    bt->max_items = max_queries + bt->target_items + (bt->use_dense ? std::max(256u, bt->tune.dense_items) : 0u);
    // scan_win_kernel's items are one wave's work each (a few per query): room for them
    if (bt->use_range && k <= scan_win_max_k(1) && bt->tune.win && !ix->term_win_host.empty())
        bt->max_items = std::max(bt->max_items, std::max(8192u, 8u * max_queries));
    int rc = 0;
    if (k > 1024) {  // exhaustive path: query buffers, results and an accumulator per document
        bt->bigk = true;
        bt->use_range = bt->use_dense = false;
        const size_t n = ix->n_docs ? ix->n_docs : 1;
        (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bt->bk_tmp_bytes, (const unsigned long long *)nullptr,
                                                     (unsigned long long *)nullptr, (const uint32_t *)nullptr,
                                                     (uint32_t *)nullptr, (int)n);
        if ((rc = bt->hits.alloc(sizeof(vbm25_hit) * size_t(max_queries) * k)) || (rc = bt->n_hits.alloc(4ull * max_queries)) ||
            (rc = bt->error_flag.alloc(4)) || (rc = bt->bk_acc.alloc(8 * n)) || (rc = bt->bk_keys.alloc(8 * n)) ||
            (rc = bt->bk_iota.alloc(4 * n)) || (rc = bt->bk_docs.alloc(4 * n)) || (rc = bt->bk_tmp.alloc(bt->bk_tmp_bytes)))
            return rc;
        HIP_TRY(hipMemset(bt->error_flag.p, 0, 4));
        bigk_iota_kernel<<<1024, 256>>>(bt->bk_iota.as<uint32_t>(), (uint32_t)n);
        HIP_TRY(hipGetLastError());
        *out = bt.release();
        return VBM25_OK;
    }
    if ((rc = bt->term_ids.alloc(4ull * max_total_terms)) ||
        (rc = bt->q_off.alloc(4ull * (max_queries + 1))) ||
        (rc = bt->items.alloc(sizeof(Item) * size_t(bt->max_items))) || (rc = bt->n_items.alloc(4)) ||
        (rc = bt->q_item_base.alloc(4ull * (max_queries + 1))) ||
        (rc = bt->theta.alloc(8ull * max_queries)) ||
        (rc = bt->res_score.alloc(8ull * bt->max_items * bt->lpi * k)) ||
        (rc = bt->res_doc.alloc(4ull * bt->max_items * bt->lpi * k)) ||
        (rc = bt->res_cnt.alloc(4ull * bt->max_items * bt->lpi)) ||
        (rc = bt->hits.alloc(sizeof(vbm25_hit) * size_t(max_queries) * k)) ||
        (rc = bt->n_hits.alloc(4ull * max_queries)) || (rc = bt->error_flag.alloc(4)) ||
        (rc = bt->q_dense.alloc(max_queries)) ||
        (rc = bt->qin.alloc(8ull * max_total_terms + 4ull * (max_queries + 1) + max_queries + 8 + 4ull * bt->max_items + 64)) ||
        (rc = bt->id16_fb.alloc(4ull * max_total_terms)) ||
        (rc = bt->item_failed.alloc(4ull * bt->max_items)) || (rc = bt->item_order.alloc(4ull * bt->max_items)) || (rc = bt->work_ctr.alloc(8)) ||
        (rc = bt->hist.alloc(4ull * CUR_HB * max_queries)) || (rc = bt->fused_state.alloc(4ull * (max_queries + 1))) ||
        (rc = bt->fail_any.alloc(4)) || (rc = bt->q_failed.alloc(4ull * max_queries)) || (rc = bt->theta_last.alloc(8ull * max_queries)))
        return rc;
    HIP_TRY(hipMemset(bt->fail_any.p, 0, 4));
    HIP_TRY(hipMemset(bt->q_failed.p, 0, 4ull * max_queries));
    HIP_TRY(hipMemset(bt->error_flag.p, 0, 4));
    // The per-launch state starts zero and every route that sets state_clean leaves ALL of it zero -- whichever route runs
    // next (the plan-free route never touches fused_state: a one-launch run after it found the allocator's leftovers there
    // and no workgroup took itself for a query's last one).
    HIP_TRY(hipMemset(bt->fused_state.p, 0, 4ull * (max_queries + 1)));
    HIP_TRY(hipMemset(bt->work_ctr.p, 0, 8));
    HIP_TRY(hipMemset(bt->theta.p, 0, 8ull * max_queries));
    HIP_TRY(hipMemset(bt->theta_last.p, 0, 8ull * max_queries));
    HIP_TRY(hipMemset(bt->hist.p, 0, 4ull * CUR_HB * max_queries));
    HIP_TRY(hipMemset(bt->item_failed.p, 0, 4ull * bt->max_items));
    HIP_TRY(hipMemset(bt->res_cnt.p, 0, 4ull * bt->max_items * bt->lpi));
    if (int rc2 = bt->dbg.alloc(64)) return rc2;
    HIP_TRY(hipMemset(bt->dbg.p, 0, 64));
#ifdef VBM25_PROFILE
    if (int rc2 = bt->prof.alloc(8ull * 16 * RNW * R_GRID)) return rc2;  // (scan_win_kernel: 768 x 4 waves -- fewer)
    HIP_TRY(hipMemset(bt->prof.p, 0, 8ull * 16 * RNW * R_GRID));
#endif
    *out = bt.release();
    return VBM25_OK;
}


void vbm25_batch_destroy(vbm25_batch *bt) {
    if (!bt) return;
    (void)hipSetDevice(bt->device);
    delete bt;
}

// queries staged in the pinned buffer (term ids | offsets | dense flags) -> device, on the batch's own stream
// The staged descriptors -- term ids, offsets, dense flags, the host's item order -- lie in ONE pinned block and go to ONE device
// block with one copy command (round 6: they were four commands into four buffers, 15 .. 25 us of a device's 100 us of host time per
// step on the multi-device route; vbm25_batch_run_impl points the kernels at the block's parts).
static int upload_staged(vbm25_batch *bt) {
    const size_t nt = bt->pin_nt, no = 4ull * (bt->nq + 1), nd8 = (size_t(bt->nq) + 7) & ~size_t(7);
    const size_t total = nt + no + nd8 + bt->pin_order_bytes + bt->pin_id16_bytes;
    if (total > bt->qin.bytes) return set_error(VBM25_ERR_INVALID, "internal error: staged descriptors exceed the device block");
    HIP_TRY(hipMemcpyAsync(bt->qin.p, bt->pin_in, total, hipMemcpyHostToDevice, bt->lat_stream));
    bt->qin_live = true;
    return VBM25_OK;
}

static int vbm25_batch_set_queries_impl(vbm25_batch *bt, const uint32_t *term_ids, const uint32_t *q_off,
                            uint32_t nq, bool fast = false) {
    if (!bt || !q_off) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (!term_ids && nq && q_off[nq] != 0) return set_error(VBM25_ERR_INVALID, "term_ids is NULL but the queries have terms");
    if (nq > bt->max_queries) return set_error(VBM25_ERR_INVALID, "%u queries exceed the batch capacity %u", nq, bt->max_queries);
    if (q_off[0] != 0) return set_error(VBM25_ERR_INVALID, "q_off[0] must be 0");
    bool many = false, has_dense = false;
    uint32_t n_dense = 0, range_mt = 0;
    // Routing: scan_range_kernel is built for sparse queries; a query with many postings per document (Zipf head terms)
    // takes the dense-window kernel, one with more than 16 indexed terms scan_many_kernel.  (The scratch vectors were
    // sized when the batch was created: nothing is allocated here.)
    const unsigned long long dense_x1000 = (unsigned long long)std::max(0ll, bt->tune.dense_x1000);
    uint8_t *dense = bt->h_dense.data();
    unsigned long long *q_postings = bt->h_postings.data();
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
        q_postings[q] = postings;
        dense[q] = 0;
        if (postings * 1000ull >= dense_x1000 * bt->index->n_docs) {
            dense[q] = 1;
            many = true;
            n_dense += postings != 0;
        }
        if (bt->use_range) {  // sparse queries of <= 16 terms: scan_range_kernel; dense ones: scan_dense_kernel; the rest: scan_many_kernel
            many |= valid > 16u;
            has_dense |= bt->use_dense && dense[q] && valid <= (uint32_t)D_T;
            if (!dense[q] && valid <= 16u) range_mt = std::max(range_mt, valid);
        } else {
            many = true;
        }
        if (valid > MAX_TERMS)
            return set_error(VBM25_ERR_UNSUPPORTED, "query %u has %u indexed terms; the GPU path handles up to %d", q, valid, MAX_TERMS);
    }
    if (q_off[nq] > bt->max_terms) return set_error(VBM25_ERR_INVALID, "%u terms exceed the batch capacity %u", q_off[nq], bt->max_terms);
    if (int rc = use_device(bt->index->device)) return rc;
    if (bt->bigk) {
        bt->h_terms.assign(term_ids, term_ids + q_off[nq]);
        bt->h_off.assign(q_off, q_off + nq + 1);
        bt->nq = nq;
        return VBM25_OK;
    }
    bt->nq = nq;
    bt->fused_g = 0;
    bt->arith_g = 0;
    bt->win_g = 0;
    bt->win_len = 0;
    bt->win_nofuse = false;
    bt->id16_decode = false;
    bt->pin_id16_bytes = 0;
    bt->n_term_pos = q_off[nq];
    bt->need_many = many || !bt->use_range;
    bt->fused_pinned = false;
    if (bt->use_range && nq && !many && !has_dense && range_mt != 0) {  // every query sparse, <= 16 indexed terms: the one-launch route
        unsigned long long most = 0;
        bool all = true;
        for (uint32_t q = 0; q < nq; ++q) {
            most = std::max(most, q_postings[q]);
            all = all && q_postings[q] != 0 && !dense[q];
        }
        if (all) {
            // items per query: by the largest query's postings, within the batch's target (every query gets the same number).
            // A handful of queries (the shim's nq = 1) is cut four times finer: its workgroups have the device to themselves,
            // and a second document range in flight is worth more than the merge of its lists costs (C2: 39.8 -> 36.2 us)
            const unsigned long long chunk = nq <= 8 ? std::max<unsigned long long>(bt->min_chunk / 4, 1) : bt->min_chunk;
            unsigned long long g = (most + chunk / 2) / chunk;
            g = std::min<unsigned long long>(g, std::max<unsigned long long>((bt->target_items + nq / 2) / nq, 1));
            g = std::min<unsigned long long>(std::max<unsigned long long>(g, 1), std::min<unsigned long long>(64, bt->index->n_docs));
            // only where the launches it saves matter: a batch that fills the GPU runs slower through the FUSED
            // instantiation (more live state in the tile loop) than plan + scan + merge cost
            // scan_win_kernel (the window formulation): every query of <= 8 indexed terms, all with a window table, the lists of
            // comparable length (the MaxScore split of scan_range_kernel has nothing to skip) and neither too thin nor too thick
            // per 2^16-document window (32 .. 232 postings on average: one 8-byte load per lane holds a run)
            uint32_t win_g = 0;
            const vbm25_index *ixh = bt->index;
            if (bt->tune.win && bt->k <= scan_win_max_k(range_mt) && !ixh->term_win_host.empty()) {
                const double wins = std::max(1.0, double(ixh->n_docs) / 65536.0);
                double e_max = 1.0;
                bool ok = true;
                for (uint32_t q = 0; q < nq && ok; ++q) {
                    uint64_t dfs[8];
                    uint32_t n = 0;
                    ok = q_off[q + 1] - q_off[q] <= 64u;
                    for (uint32_t p = q_off[q]; p < q_off[q + 1] && ok; ++p) {
                        const uint32_t t = term_ids[p];
                        if (t >= ixh->n_terms) continue;
                        ok = n < scan_win_max_terms() && ixh->term_win_host[t] != UINT32_MAX;
                        if (!ok) break;
                        const double e = double(ixh->term_df_host[t]) / wins;
                        ok = bt->tune.win_force || (e >= 32.0 && e <= 232.0);
                        e_max = std::max(e_max, e);
                        dfs[n++] = ixh->term_df_host[t];
                    }
                    if (ok && !bt->tune.win_force && bt->tune.ne) {  // a prefix of the longest lists that scan_range_kernel would look up instead of scanning?
                        std::sort(dfs, dfs + n, [](uint64_t a, uint64_t b) { return a > b; });
                        uint64_t rest = 0;
                        for (uint32_t i = 0; i < n; ++i) rest += dfs[i];
                        for (uint32_t i = 0; i + 1 < n && ok; ++i) {
                            rest -= dfs[i];
                            ok = !(dfs[i] >= uint64_t(std::max(1u, bt->tune.ne_ratio)) * rest || dfs[i] * 16u >= ixh->n_docs);
                        }
                    }
                }
                if (ok) {
                    // items: one per resident wave (an item's setup is a chain of six round trips to memory: on C3 3072 items of 51
                    // windows take 0.270 ms, 6144 of 25 windows 0.287 ms)
                    // ... of as many windows as give every resident wave the same share of the batch's windows: L = ceil(nq n_win / waves).
                    // A query is cut into runs of L windows and a shorter rest (win_cut, vbm25_batch_run): with 14 waves per workgroup C3's
                    // 1024 queries x 153 windows are 3072 runs of 44 and 1024 of 21 for 3584 waves -- the short ones go last, two to a wave.
                    const uint32_t target = bt->tune.win_items ? bt->tune.win_items : scan_win_resident_waves(range_mt, bt->k);
                    const uint32_t g_min = (ixh->n_win + 62u) / 63u;  // (an item holds at most 63 windows)
                    const uint64_t all_win = uint64_t(nq) * ixh->n_win;
                    const uint32_t len = uint32_t(std::min<uint64_t>(63u, std::max<uint64_t>(1u, (all_win + target - 1u) / target)));
                    uint32_t gw = std::max((ixh->n_win + len - 1u) / len, g_min);
                    gw = std::min(std::min(gw, ixh->n_win), bt->max_items / nq);
                    if (gw >= g_min && gw >= 1u) {
                        win_g = gw;
                        bt->win_len = (ixh->n_win + gw - 1u) / gw > len ? 0u : len;  // (0: equal runs -- the item limit cut the number of runs)
                    }
                }
            }
            if (bt->tune.fused && nq * g <= bt->tune.fused_items) {
                bt->fused_g = uint32_t(g);
                bt->fused_pinned = fast && nq <= 8 && !bt->timing;
            } else if (win_g || bt->tune.arith) {
                if (win_g) {
                    bt->win_g = win_g;
                    g = win_g;
                    if (!ixh->post_id16.p) {
                        // An index without the post_id16 plane: every term position gets its blocks in the batch's scratch plane (a
                        // term's blocks are full but its last: (df + 127) / 128 of them, post_fn_kernel's flag 8 saw to that), which
                        // decode_id16_kernel fills ahead of the scan.  The plane grows with the largest batch it has held.
                        std::vector<uint32_t> &fbv = bt->h_id16_fb;
                        fbv.resize(q_off[nq]);
                        uint64_t blocks = 2;  // (block 0: what the null terms' run loads read)
                        for (uint32_t p = 0; p < q_off[nq]; ++p) {
                            const uint32_t t = term_ids[p];
                            fbv[p] = uint32_t(blocks);
                            if (t < ixh->n_terms) blocks += (uint64_t(ixh->term_df_host[t]) + 127u) / 128u;
                        }
                        if (blocks > 0x00ffffffull) return set_error(VBM25_ERR_UNSUPPORTED, "the batch's terms exceed the scratch plane of an index without post_id16");
                        const size_t need = 256ull * blocks + 1024;
                        if (need > bt->id16_tmp.bytes) {
                            if (bt->last_stream || bt->lat_stream) HIP_TRY(hipDeviceSynchronize());  // (a run still reading the old plane)
                            if (bt->id16_tmp.p) HIP_TRY(hipFree(bt->id16_tmp.p));
                            bt->id16_tmp.p = nullptr;
                            bt->id16_tmp.bytes = 0;
                            if (int rc = bt->id16_tmp.alloc(need + need / 4)) return rc;
                            HIP_TRY(hipMemset(bt->id16_tmp.p, 0, bt->id16_tmp.bytes));
                        }
                        bt->id16_decode = true;
                        if (!fast && q_off[nq]) HIP_TRY(hipMemcpy(bt->id16_fb.p, fbv.data(), 4ull * q_off[nq], hipMemcpyHostToDevice));
                    }
                } else
                bt->arith_g = uint32_t(g);  // the general route, items made in the kernel: no plan_kernel, merge_kernel cleans
                // ... handed out longest first, as plan_kernel would (the host has the posting counts): queries by
                // descending postings, a query's g parts together
                std::vector<uint32_t> &ord = bt->h_order;
                ord.resize(size_t(nq) * g);
                std::vector<uint32_t> &qs = bt->h_order_q;
                qs.resize(nq);
                for (uint32_t q = 0; q < nq; ++q) qs[q] = q;
                {   // (queries of about the same length -- C3's -- keep their order: the sort, with the scratch buffer std::stable_sort
                    // allocates, was a quarter of a device's host time per step on the multi-device route)
                    unsigned long long lo = ~0ull, hi = 0;
                    for (uint32_t q = 0; q < nq; ++q) {
                        lo = std::min(lo, q_postings[q]);
                        hi = std::max(hi, q_postings[q]);
                    }
                    bt->order_identity = !(hi * 4 > lo * 5);  // (the queries keep their order: the skewed layout is arithmetic -- scan_win.h)
                    if (hi * 4 > lo * 5) std::stable_sort(qs.begin(), qs.end(), [&](uint32_t a, uint32_t b) { return q_postings[a] > q_postings[b]; });
                }

                const uint32_t wpw = win_g ? scan_win_wg(range_mt, bt->k) : 0u;
                bt->win_skew = win_g == 3u && wpw == 12u && size_t(nq) * 3u <= scan_win_resident_waves(range_mt, bt->k) && bt->tune.win_skew;
                // (queries of about the same length: the longest-first order buys nothing and costs every work item a dependent load --
                // unless there are more items than waves: then the queries' short last runs must be the ones drawn late)
                bt->order_useful = bt->win_skew || (!win_g || !bt->tune.win_guided ? q_postings[qs[0]] * 4 > q_postings[qs[nq - 1]] * 5 : true) ||
                                   (win_g && size_t(nq) * win_g > scan_win_resident_waves(range_mt, bt->k));
                if (bt->win_skew) {
                    // One item per wave, three per query: a SIMD's three waves do not run equally fast -- the workgroup's waves 0..3
                    // (the first wave of every SIMD) lived 449 k cycles on C3, 4..7 497 k, 8..11 559 k, whatever priority they set
                    // themselves -- and the launch ends with the slowest.  A workgroup takes four queries; a query's three runs of
                    // windows, of lengths in the ratio of those speeds (win_cut, vbm25_batch_run), go to one wave of each kind.
                    // (nq mod 4 queries are left over: their items follow in plain order -- written to a partial last workgroup's
                    // slots they would land beyond the end of the array)
                    const uint32_t full = nq / 4u;
                    for (uint32_t wgi = 0; wgi < full; ++wgi)
                        for (uint32_t s4 = 0; s4 < 4u; ++s4)
                            for (uint32_t part = 0; part < 3u; ++part) ord[size_t(wgi) * 12u + part * 4u + s4] = qs[wgi * 4u + s4] * 3u + part;
                    for (uint32_t qi = full * 4u; qi < nq; ++qi)
                        for (uint32_t part = 0; part < 3u; ++part) ord[size_t(qi) * 3u + part] = qs[qi] * 3u + part;
                } else if (win_g)  // (parts of decreasing length: every query's first part, then every query's second one, ...)
                    for (uint32_t part = 0; part < g; ++part)
                        for (uint32_t i = 0; i < nq; ++i) ord[size_t(part) * nq + i] = qs[i] * uint32_t(g) + part;
                else
                for (uint32_t i = 0; i < nq; ++i)
                    for (uint32_t part = 0; part < g; ++part) ord[size_t(i) * g + part] = qs[i] * uint32_t(g) + part;
                // (the fast path stages the order with the queries and copies it on the batch's own stream: upload_staged)
                if (!fast) HIP_TRY(hipMemcpy(bt->item_order.p, ord.data(), 4ull * ord.size(), hipMemcpyHostToDevice));
            }
        }
    }
    if (fast && !bt->bigk) {
        // staged in pinned memory.  One-launch route (fused_g): the kernel reads the queries from there and writes the hits
        // into the pinned output buffer -- no copy is enqueued at all.  General route: copied on the batch's own stream,
        // nothing waits here.
        const size_t nt = 4ull * q_off[nq], no = 4ull * (nq + 1);
        const bool with_order = bt->arith_g || bt->win_g;
        const size_t nd8 = (size_t(nq) + 7) & ~size_t(7), nord = with_order ? 4ull * bt->h_order.size() : 0;
        const size_t nid = bt->id16_decode ? nt : 0;  // (the terms' blocks in the scratch plane of an index without post_id16)
        if (nt + no + nd8 + nord + nid > bt->pin_in_bytes) {
            if (bt->pin_in) HIP_TRY(hipHostFree(bt->pin_in));
            bt->pin_in = nullptr;
            bt->pin_in_bytes = 2 * (nt + no + nd8 + nord + nid) + 256;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&bt->pin_in), bt->pin_in_bytes, hipHostMallocDefault));
        }
        const size_t nh = sizeof(vbm25_hit) * size_t(nq) * bt->k, nc = (4ull * nq + 7) & ~size_t(7);
        if (8 + nc + nh > bt->pin_out_bytes) {
            if (bt->pin_out) HIP_TRY(hipHostFree(bt->pin_out));
            bt->pin_out = nullptr;
            bt->pin_out_bytes = 2 * (8 + nc + nh) + 256;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&bt->pin_out), bt->pin_out_bytes, hipHostMallocDefault));
        }
        if (!bt->lat_stream) HIP_TRY(hipStreamCreateWithFlags(&bt->lat_stream, hipStreamNonBlocking));
        if (nt) std::memcpy(bt->pin_in, term_ids, nt);
        std::memcpy(bt->pin_in + nt, q_off, no);
        if (nq) std::memcpy(bt->pin_in + nt + no, dense, nq);
        if (nord) std::memcpy(bt->pin_in + nt + no + nd8, bt->h_order.data(), nord);
        if (nid) std::memcpy(bt->pin_in + nt + no + nd8 + nord, bt->h_id16_fb.data(), nid);
        bt->pin_nt = nt;
        bt->pin_order_bytes = nord;
        bt->pin_id16_bytes = nid;
        if (!(bt->fused_g && bt->fused_pinned))
            if (int rc = upload_staged(bt)) return rc;
    } else {
        bt->qin_live = false;
        if (q_off[nq]) HIP_TRY(hipMemcpy(bt->term_ids.p, term_ids, 4ull * q_off[nq], hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(bt->q_off.p, q_off, 4ull * (nq + 1), hipMemcpyHostToDevice));
        if (nq) HIP_TRY(hipMemcpy(bt->q_dense.p, dense, nq, hipMemcpyHostToDevice));
    }
    bt->has_dense = has_dense;
    bt->range_rt = !bt->use_range || range_mt == 0 ? 0u : (range_mt <= 8u ? 8u : 16u);
    bt->win_mt = range_mt;
    bt->q_stride = 0;
    if (nq && q_off[1] != 0) {
        bt->q_stride = q_off[1];
        for (uint32_t q = 0; q < nq && bt->q_stride; ++q)
            if (q_off[q + 1] - q_off[q] != bt->q_stride) bt->q_stride = 0;
    }
    {   // the number of work items plan_kernel will make (same integer arithmetic): the persistent grids need not be
        // larger (a single query is a handful of items); with the dense-window kernel every dense query gets the same
        // number of items (equal document counts)
        const uint32_t dense_target = std::max(256u, bt->tune.dense_items);
        bt->dense_c = has_dense ? std::max(1u, (dense_target + n_dense / 2) / std::max(n_dense, 1u)) : 0u;
        if (has_dense) {
            // (a few dense queries on a small corpus: no finer than 4096 items in all -- what the rounds before used -- or one item per
            // 2^16 documents, whichever is more: below that an item is all setup)
            const uint32_t floor_c = std::max(std::max(1u, 4096u / std::max(n_dense, 1u)), bt->index->n_docs >> 16);
            if (bt->tune.dense_items == D_TARGET_ITEMS) bt->dense_c = std::min(bt->dense_c, floor_c);
        }
        unsigned long long sparse_postings = 0;
        for (uint32_t q = 0; q < nq; ++q)
            if (!(bt->dense_c && dense[q])) sparse_postings += q_postings[q];
        unsigned long long chunk = (sparse_postings + bt->target_items - 1) / bt->target_items;
        if (chunk < bt->min_chunk) chunk = bt->min_chunk;
        unsigned long long items = 0;
        for (uint32_t q = 0; q < nq; ++q) {
            if (!q_postings[q]) continue;
            if (bt->dense_c && dense[q]) {
                items += std::min(bt->dense_c, bt->index->n_docs);
                continue;
            }
            unsigned long long c = (q_postings[q] + chunk / 2) / chunk;
            if (c == 0) c = 1;
            if (c > bt->index->n_docs) c = bt->index->n_docs;
            items += c;
        }
        bt->range_grid = uint32_t(std::min<unsigned long long>(std::max<unsigned long long>(items, 1), std::max(1u, bt->tune.range_grid)));
        bt->dense_grid = uint32_t(std::min<unsigned long long>(std::max<unsigned long long>(items, 1), std::max(1u, bt->tune.dense_grid)));
#ifdef VBM25_PROFILE
        bt->range_grid = std::min<uint32_t>(bt->range_grid, R_GRID);  // (the phase counters are sized for R_GRID workgroups)
#endif
    }
    return VBM25_OK;
}


static int vbm25_batch_run_impl(vbm25_batch *bt, void *hip_stream) {
    if (!bt) return set_error(VBM25_ERR_INVALID, "batch is NULL");
    if (!bt->nq) return VBM25_OK;
    if (int rc = use_device(bt->index->device)) return rc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    bt->last_stream = st;
    bt->win_fused_run = false;
    bt->download_enqueued = false;
    bt->results_pinned_now = false;
    if (bt->index->n_docs == 0) {  // empty sealed segment: no hits (the growing segment is the shim's, search.rs:83-135)
        HIP_TRY(hipMemsetAsync(bt->n_hits.p, 0, 4ull * bt->nq, st));
        return VBM25_OK;
    }
    if (bt->bigk) {
        const DevIndex &dix = bt->index->dev;
        const uint32_t n = bt->index->n_docs;
        for (uint32_t q = 0; q < bt->nq; ++q) {
            HIP_TRY(hipMemsetAsync(bt->bk_acc.p, 0, 8ull * n, st));
            for (uint32_t p = bt->h_off[q]; p < bt->h_off[q + 1]; ++p) {
                const uint32_t t = bt->h_terms[p];
                if (t >= bt->index->n_terms) continue;  // search.rs:59-61
                const uint32_t nb = (bt->index->term_df_host[t] + 127) / 128;
                bigk_accum_kernel<<<std::min<uint32_t>((nb + 3) / 4, 4096u), 256, 0, st>>>(dix, t, bt->bk_acc.as<double>());
            }
            size_t tmp = bt->bk_tmp_bytes;
            HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(bt->bk_tmp.p, tmp, bt->bk_acc.as<unsigned long long>(),
                                                                  bt->bk_keys.as<unsigned long long>(), bt->bk_iota.as<uint32_t>(),
                                                                  bt->bk_docs.as<uint32_t>(), (int)n, 0, 64, st));
            bigk_emit_kernel<<<(bt->k + 255) / 256, 256, 0, st>>>(dix, bt->bk_keys.as<unsigned long long>(), bt->bk_docs.as<uint32_t>(), n,
                                                                  bt->k, bt->hits.as<vbm25_hit>() + size_t(q) * bt->k,
                                                                  bt->n_hits.as<uint32_t>() + q);
        }
        HIP_TRY(hipGetLastError());
        return VBM25_OK;
    }
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
    bt->results_pinned_now = false;
    if (bt->pinned_results && !bt->bigk && !bt->fused_g && bt->lat_stream && bt->pin_out) {
        const size_t nc = (4ull * bt->nq + 7) & ~size_t(7);  // (set_queries sized pin_out for 8 + nc + the records)
        db.n_hits = reinterpret_cast<uint32_t *>(bt->pin_out + 8);
        db.hits = reinterpret_cast<vbm25_hit *>(bt->pin_out + 8 + nc);
        bt->results_pinned_now = true;
    }
    db.error_flag = bt->error_flag.as<uint32_t>();
    db.q_dense = bt->q_dense.as<uint8_t>();
    if (bt->qin_live) {  // the staged descriptors: one device block (upload_staged)
        uint8_t *qp = bt->qin.as<uint8_t>();
        const size_t nt = bt->pin_nt, no = 4ull * (bt->nq + 1), nd8 = (size_t(bt->nq) + 7) & ~size_t(7);
        db.term_ids = reinterpret_cast<const uint32_t *>(qp);
        db.q_off = reinterpret_cast<const uint32_t *>(qp + nt);
        db.q_dense = qp + nt + no;
    }
    db.item_failed = bt->item_failed.as<uint32_t>();
    db.item_order = bt->item_order.as<uint32_t>();
    if (bt->qin_live && bt->pin_order_bytes)  // (the host's item order of the routes without plan_kernel, which writes its own into item_order)
        db.item_order = reinterpret_cast<uint32_t *>(bt->qin.as<uint8_t>() + bt->pin_nt + 4ull * (bt->nq + 1) + ((size_t(bt->nq) + 7) & ~size_t(7)));
    db.prof = bt->prof.as<unsigned long long>();
    db.hist = bt->hist.as<uint32_t>();
    db.work_ctr = bt->work_ctr.as<uint32_t>();
    db.max_items = bt->max_items;
    db.dbg = bt->dbg.as<uint32_t>();
    db.lpi = bt->lpi;
    db.range_max_terms = bt->use_range ? 16u : 0u;
    db.ne_on = bt->tune.ne ? 1u : 0u;
    db.ne_ratio = std::max(1u, bt->tune.ne_ratio);
    db.dense_on = bt->use_dense ? 1u : 0u;
    db.win_dbg = bt->tune.dbg;
    db.fail_any = bt->fail_any.as<uint32_t>();
    db.q_failed = bt->q_failed.as<uint32_t>();
    db.theta_last = bt->theta_last.as<unsigned long long>();
    db.many_expected = bt->need_many ? 1u : 0u;
    db.merge_clean = 0;
    db.order_on = 0;
    const bool range = bt->use_range;
    const DevIndex &ix = bt->index->dev;
    db.fused_state = bt->fused_state.as<uint32_t>();
    db.fused_g = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto take_events = [&]() -> int {
        if (!bt->timing) return VBM25_OK;
        if (bt->events_used == bt->events.size()) {
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            bt->events.emplace_back(e0, e1);
        }
        e0 = bt->events[bt->events_used].first;
        e1 = bt->events[bt->events_used].second;
        bt->events_used++;
        HIP_TRY(hipEventRecord(e0, st));
        return VBM25_OK;
    };
    if (bt->fused_g && bt->range_rt) {
        // The one-launch route needs threshold, histogram and counters zero; it leaves them so itself -- but only a run
        // that was enqueued completely counts: the flag is cleared before anything is enqueued and set at the very end
        // (an error return in between, or a run on the general route, forces the memsets next time).  Consecutive runs
        // of one batch must use one stream.
        const bool clean = bt->state_clean;
        bt->state_clean = false;
        if (!clean) {
            HIP_TRY(hipMemsetAsync(bt->hist.p, 0, 4ull * CUR_HB * bt->max_queries, st));
            HIP_TRY(hipMemsetAsync(bt->theta.p, 0, 8ull * bt->max_queries, st));
            HIP_TRY(hipMemsetAsync(bt->work_ctr.p, 0, 8, st));
            HIP_TRY(hipMemsetAsync(bt->fused_state.p, 0, 4ull * (bt->max_queries + 1), st));
        }
        db.fused_g = bt->fused_g;
        db.dense_on = 0;
        if (bt->fused_pinned) {  // queries read from, hits written to pinned host memory
            const size_t nc = (4ull * bt->nq + 7) & ~size_t(7);
            db.term_ids = reinterpret_cast<const uint32_t *>(bt->pin_in);
            db.q_off = reinterpret_cast<const uint32_t *>(bt->pin_in + bt->pin_nt);
            db.n_hits = reinterpret_cast<uint32_t *>(bt->pin_out + 8);
            db.hits = reinterpret_cast<vbm25_hit *>(bt->pin_out + 8 + nc);
        }
        if (int rc = take_events()) return rc;
        const uint32_t fgrid = std::min<uint32_t>(bt->nq * bt->fused_g, R_GRID);
        const int rcf = dispatch_k(bt->k, [&](auto kmax) {
            constexpr int KM = decltype(kmax)::value;
            if constexpr (KM <= REG_K) {
                if (bt->range_rt == 8) scan_range_kernel<KM, 8, true><<<fgrid, RWG, 0, st>>>(ix, db);
                else scan_range_kernel<KM, 16, true><<<fgrid, RWG, 0, st>>>(ix, db);
                if (!bt->fused_pinned) {
                    // an item the kernel gave up (rare) is redone by scan_many_kernel and its query merged by merge_kernel;
                    // both find nothing to do otherwise.  (The pinned flavour leaves that to the host: it re-runs the batch.)
                    scan_many_kernel<KM><<<std::min<uint32_t>(bt->nq * bt->fused_g, TARGET_ITEMS), WG, 0, st>>>(ix, db);
                    if (bt->timing) (void)hipEventRecord(e1, st);
                    DevBatch dm = db;
                    dm.merge_marked = 1;
                    merge_kernel<KM><<<bt->nq, 64, 0, st>>>(ix, dm);
                } else if (bt->timing) {
                    (void)hipEventRecord(e1, st);
                }
            }
            return int(VBM25_OK);
        });
        if (rcf) return rcf;
        HIP_TRY(hipGetLastError());
        bt->state_clean = true;
        return VBM25_OK;
    }
    if (bt->win_g && bt->k <= scan_win_max_k(bt->win_mt)) {
        // Every query sparse, <= 8 terms of comparable length: scan_win_kernel (its waves make their work items themselves), then as
        // on the route below: scan_many_kernel leaves at once unless an item was given up, merge_kernel merges and cleans.
        const bool clean = bt->state_clean;
        bt->state_clean = false;
        if (!clean) {
            HIP_TRY(hipMemsetAsync(bt->hist.p, 0, 4ull * CUR_HB * bt->max_queries, st));
            HIP_TRY(hipMemsetAsync(bt->theta.p, 0, 8ull * bt->max_queries, st));
            HIP_TRY(hipMemsetAsync(bt->work_ctr.p, 0, 8, st));
            HIP_TRY(hipMemsetAsync(bt->fused_state.p, 0, 4ull * (bt->max_queries + 1), st));
            HIP_TRY(hipMemsetAsync(bt->fail_any.p, 0, 4, st));
            HIP_TRY(hipMemsetAsync(bt->item_failed.p, 0, 4ull * bt->max_items, st));
            HIP_TRY(hipMemsetAsync(bt->res_cnt.p, 0, 4ull * bt->max_items * bt->lpi, st));
        }
        db.fused_g = bt->win_g;  // (merge_kernel: a query's lists are those of its win_g items)
        db.win_g = bt->win_g;
        {
            // a query's items: runs of windows of decreasing length (weights g + 1, g, ..., 2), handed out longest first -- the
            // last items drawn, which decide when the launch ends, are the short ones.  Equal runs when a run would exceed the
            // 63 windows an item can hold, or with more than 16 items per query.
            const uint32_t gq = bt->win_g, nwin = bt->index->n_win;
            std::memset(db.win_cut, 0, sizeof db.win_cut);
            if (bt->win_skew) {  // (the three kinds of waves: 59 / 52 / 42 of C3's 153 windows: the third wave of a SIMD takes 1.5 times the first one's time per window)
                db.win_cut[0] = 0;
                db.win_cut[1] = uint32_t(uint64_t(nwin) * uint32_t(bt->tune.win_cut1) / 1000u);
                db.win_cut[2] = uint32_t(uint64_t(nwin) * uint32_t(bt->tune.win_cut2) / 1000u);
                db.win_cut[3] = nwin;
                if (db.win_cut[1] > 63u || db.win_cut[2] - db.win_cut[1] > 63u || nwin - db.win_cut[2] > 63u) std::memset(db.win_cut, 0, sizeof db.win_cut);
            } else if (gq <= 16u && bt->win_len && uint64_t(bt->win_len) * (gq - 1u) < nwin) {
                // runs of win_len windows and a shorter rest (set_queries: every resident wave the same share of the windows)
                for (uint32_t p = 0; p < gq; ++p) db.win_cut[p] = bt->win_len * p;
                db.win_cut[gq] = nwin;
            }
        }
        db.lpi = 1;
        db.hist = nullptr;       // (this kernel keeps no histogram of accepted documents: merge_kernel has none to clean)
        db.dense_on = 0;
        db.many_expected = 0;
        db.merge_clean = 1;
        // (2: the skewed layout of queries in the caller's order is computed by the kernel itself, nobody loads item_order)
        db.order_on = bt->order_useful ? (bt->win_skew && bt->order_identity && bt->tune.win_order_arith ? 2u : 1u) : 0u;
        db.q_stride = bt->q_stride;
        if (int rc = take_events()) return rc;
        const uint32_t wmt = bt->range_rt == 8 ? bt->win_mt : 8u, wpw = scan_win_wg(wmt, bt->k);
        const uint32_t wgrid = std::min<uint32_t>((bt->nq * bt->win_g + wpw - 1u) / wpw, bt->tune.win_grid ? bt->tune.win_grid : scan_win_resident_waves(wmt, bt->k) / wpw);
        // One launch (round 6): the wave that finishes a query's last item merges the query's lists, writes its records and leaves
        // the per-launch state zero.  A query with an item the kernel gave up comes back with the count NONE32: whoever hands the
        // records to the caller (vbm25_batch_fetch_impl) re-runs the batch with scan_many_kernel and merge_kernel behind the scan.
        // (only when every wave has at most one item: the kernel's merge sits behind its item loop)
        const bool fuse = bt->tune.win_fuse && !bt->win_nofuse && !bt->device_consumer && uint64_t(bt->nq) * bt->win_g <= scan_win_resident_waves(wmt, bt->k) && !bt->tune.win_grid;
        db.win_fuse = fuse ? 1u : 0u;
        bt->win_fused_run = fuse;
        if (bt->id16_decode) {
            // An index without the post_id16 plane: the ids of the batch's terms, unpacked from the blob into the batch's scratch
            // plane (decode_id16.h) -- inside the timed region: kernel_ms is the decode and the scan.
            db.id16_fb = bt->id16_fb.as<uint32_t>();
            if (bt->qin_live && bt->pin_id16_bytes)
                db.id16_fb = reinterpret_cast<const uint32_t *>(bt->qin.as<uint8_t>() + bt->pin_nt + 4ull * (bt->nq + 1) + ((size_t(bt->nq) + 7) & ~size_t(7)) + bt->pin_order_bytes);
            const uint32_t n_pos = bt->n_term_pos;
            if (n_pos) decode_id16_kernel<<<n_pos, DI_WAVES * 64, 0, st>>>(ix, db.term_ids, db.id16_fb, bt->id16_tmp.as<uint32_t>());
            DevIndex ixw = ix;
            ixw.post_id16 = bt->id16_tmp.as<uint32_t>();
            HIP_TRY(scan_win_launch(ixw, db, wmt, wgrid, st));
        } else
        HIP_TRY(scan_win_launch(ix, db, wmt, wgrid, st));
        if (fuse) {
            if (bt->timing) (void)hipEventRecord(e1, st);
        } else
        (void)dispatch_k(bt->k, [&](auto kmax) {
            constexpr int KM = decltype(kmax)::value;
            if constexpr (KM <= REG_K) {
                scan_many_kernel<KM><<<64, WG, 0, st>>>(ix, db);
                if (bt->timing) (void)hipEventRecord(e1, st);
                merge_kernel<KM><<<bt->nq, 64, 0, st>>>(ix, db);
            }
            return int(VBM25_OK);
        });
        HIP_TRY(hipGetLastError());
        bt->state_clean = true;
        return VBM25_OK;
    }
    if (bt->arith_g && bt->range_rt && bt->k <= (uint32_t)REG_K) {
        // Every query sparse, <= 16 terms: the scan kernel makes the work items itself (no plan_kernel), scan_many_kernel is a
        // small grid that leaves at once unless an item was given up, merge_kernel merges and leaves the per-launch state zero.
        const bool clean = bt->state_clean;
        bt->state_clean = false;
        if (!clean) {
            HIP_TRY(hipMemsetAsync(bt->hist.p, 0, 4ull * CUR_HB * bt->max_queries, st));
            HIP_TRY(hipMemsetAsync(bt->theta.p, 0, 8ull * bt->max_queries, st));
            HIP_TRY(hipMemsetAsync(bt->work_ctr.p, 0, 8, st));
            HIP_TRY(hipMemsetAsync(bt->fused_state.p, 0, 4ull * (bt->max_queries + 1), st));
            HIP_TRY(hipMemsetAsync(bt->fail_any.p, 0, 4, st));
            HIP_TRY(hipMemsetAsync(bt->item_failed.p, 0, 4ull * bt->max_items, st));
            HIP_TRY(hipMemsetAsync(bt->res_cnt.p, 0, 4ull * bt->max_items * bt->lpi, st));
        }
        db.fused_g = bt->arith_g;
        db.dense_on = 0;
        db.many_expected = 0;
        db.merge_clean = 1;
        db.order_on = 1;
        if (int rc = take_events()) return rc;
        const uint32_t agrid = std::min<uint32_t>(bt->nq * bt->arith_g, std::max(1u, bt->tune.range_grid));
        const int rca = dispatch_k(bt->k, [&](auto kmax) {
            constexpr int KM = decltype(kmax)::value;
            if constexpr (KM <= REG_K) {
                if (bt->range_rt == 8) scan_range_kernel<KM, 8><<<agrid, RWG, 0, st>>>(ix, db);
                else scan_range_kernel<KM, 16><<<agrid, RWG, 0, st>>>(ix, db);
                scan_many_kernel<KM><<<64, WG, 0, st>>>(ix, db);
                if (bt->timing) (void)hipEventRecord(e1, st);
                merge_kernel<KM><<<bt->nq, 64, 0, st>>>(ix, db);
            }
            return int(VBM25_OK);
        });
        if (rca) return rca;
        HIP_TRY(hipGetLastError());
        bt->state_clean = true;
        return VBM25_OK;
    }
    bt->state_clean = false;
    if (range) HIP_TRY(hipMemsetAsync(bt->hist.p, 0, 4ull * CUR_HB * bt->nq, st));
    plan_kernel<<<1, PLAN_WG, 0, st>>>(ix, db, bt->max_items, bt->target_items, bt->min_chunk, bt->dense_c);
    if (int rc = take_events()) return rc;
    const uint32_t grid = std::min<uint32_t>(bt->max_items, TARGET_ITEMS);
    const int rc = dispatch_k(bt->k, [&](auto kmax) {
        constexpr int KM = decltype(kmax)::value;
        if constexpr (KM <= REG_K) {
            if (range) {  // persistent 8-wave workgroups
                if (bt->range_rt == 8) scan_range_kernel<KM, 8><<<bt->range_grid, RWG, 0, st>>>(ix, db);
                else if (bt->range_rt == 16) scan_range_kernel<KM, 16><<<bt->range_grid, RWG, 0, st>>>(ix, db);
            }
        }
        if constexpr (KM <= D_KMAX) {
            if (bt->has_dense) scan_dense_kernel<KM><<<bt->dense_grid, DWG, 0, st>>>(ix, db);
        }
        // many-term queries, every query of 256 < k <= 1024, and items the first-choice kernel gave up on (empty launch: 5 us)
        scan_many_kernel<decltype(kmax)::value><<<grid, WG, 0, st>>>(ix, db);
        if (bt->timing) HIP_TRY(hipEventRecord(e1, st));  // the events bracket every posting-scan kernel of the step
        merge_kernel<decltype(kmax)::value><<<bt->nq, 64, 0, st>>>(ix, db);
        return int(VBM25_OK);
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return VBM25_OK;
}

static int vbm25_batch_enqueue_download(vbm25_batch *bt);
static int vbm25_batch_fetch_impl(vbm25_batch *bt, vbm25_hit *hits, uint32_t *n_hits, bool fast = false);
// After a one-launch run of scan_win_kernel (win_fuse): a count of NONE32 marks a query with an item the kernel gave up (more second
// arrivals in a window than its list holds, a term frequency above 255).  The batch is run again with scan_many_kernel and
// merge_kernel behind the scan -- on the same stream, before any record reaches the caller -- and this query set stays on that route.
static bool win_marked(const vbm25_batch *bt, const uint32_t *cnt) {
    if (!bt->win_fused_run) return false;
    for (uint32_t q = 0; q < bt->nq; ++q)
        if (cnt[q] == UINT32_MAX) return true;
    return false;
}
static int win_rerun_and_fetch(vbm25_batch *bt, vbm25_hit *hits, uint32_t *n_hits, bool fast) {
    bt->win_nofuse = true;
    bt->state_clean = false;  // (the one-launch run left fail_any set)
    bt->download_enqueued = false;
    if (int rc = vbm25_batch_run_impl(bt, bt->last_stream)) return rc;
    return vbm25_batch_fetch_impl(bt, hits, n_hits, fast);
}
static int vbm25_batch_fetch_impl(vbm25_batch *bt, vbm25_hit *hits, uint32_t *n_hits, bool fast) {
    if (!bt || (!hits && bt->nq) || (!n_hits && bt->nq)) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (int rc = use_device(bt->index->device)) return rc;
    if (fast && bt->lat_stream && bt->fused_g && bt->fused_pinned) {  // the kernel wrote counts and hits into the pinned buffer: one synchronisation
        const size_t nh = sizeof(vbm25_hit) * size_t(bt->nq) * bt->k, nc = (4ull * bt->nq + 7) & ~size_t(7);
        HIP_TRY(hipStreamSynchronize(bt->lat_stream));
        const uint32_t *cnt = reinterpret_cast<const uint32_t *>(bt->pin_out + 8);
        for (uint32_t q = 0; q < bt->nq; ++q)
            if (cnt[q] == UINT32_MAX) return VBM25_RETRY_GENERAL;  // an item needs scan_many_kernel
        std::memcpy(n_hits, cnt, 4ull * bt->nq);
        std::memcpy(hits, bt->pin_out + 8 + nc, nh);
        return VBM25_OK;
    }
    if (fast && bt->lat_stream) {  // flag, counts and hits come down asynchronously; one synchronisation
        const size_t nh = sizeof(vbm25_hit) * size_t(bt->nq) * bt->k;
        const size_t nc = bt->results_pinned_now ? (4ull * bt->nq + 7) & ~size_t(7) : 4ull * bt->nq;  // (the kernel's records are 8-byte aligned)
        if (!bt->download_enqueued)
            if (int rc = vbm25_batch_enqueue_download(bt)) return rc;
        bt->download_enqueued = false;
        HIP_TRY(hipStreamSynchronize(bt->lat_stream));
        uint32_t flag = 0;
        std::memcpy(&flag, bt->pin_out, 4);
        if (flag) {
            HIP_TRY(hipMemset(bt->error_flag.p, 0, 4));
            return set_error(VBM25_ERR_DEVICE, "device-side planner overflow (flag %u)", flag);
        }
        if (bt->nq) {
            if (win_marked(bt, reinterpret_cast<const uint32_t *>(bt->pin_out + 8))) return win_rerun_and_fetch(bt, hits, n_hits, fast);
            // (nc is the 8-byte aligned OFFSET of the records; the caller's n_hits holds exactly nq counts: copying nc bytes wrote four
            // bytes past it for an odd nq -- into the next shard's first count on the multi-device route)
            std::memcpy(n_hits, bt->pin_out + 8, 4ull * bt->nq);
            std::memcpy(hits, bt->pin_out + 8 + nc, nh);
        }
        return VBM25_OK;
    }
    // the batch's own stream, not the device: other streams of the process (a gather, another batch) keep running
    hipStream_t st = bt->last_stream;
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, bt->error_flag.p, 4, hipMemcpyDeviceToHost, st));
    if (bt->nq) {
        HIP_TRY(hipMemcpyAsync(hits, bt->hits.p, sizeof(vbm25_hit) * size_t(bt->nq) * bt->k, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(n_hits, bt->n_hits.p, 4ull * bt->nq, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (flag) {
        HIP_TRY(hipMemsetAsync(bt->error_flag.p, 0, 4, st));
        HIP_TRY(hipStreamSynchronize(st));
        return set_error(VBM25_ERR_DEVICE, "device-side planner overflow (flag %u)", flag);
    }
    if (bt->nq && win_marked(bt, n_hits)) return win_rerun_and_fetch(bt, hits, n_hits, fast);
    return VBM25_OK;
}

// the last run's flag, counts and records to the batch's pinned buffer, behind the run on its private stream (vbm25_search_batch's
// and vbm25_multi_batch_run's general routes; the one-launch route has written them there itself)
static int vbm25_batch_enqueue_download(vbm25_batch *bt) {
    if (bt->bigk || !bt->lat_stream || (bt->fused_g && bt->fused_pinned)) return VBM25_OK;
    if (int rc = use_device(bt->index->device)) return rc;
    const size_t nh = sizeof(vbm25_hit) * size_t(bt->nq) * bt->k, nc = 4ull * bt->nq;
    if (8 + nc + nh > bt->pin_out_bytes) {
        if (bt->pin_out) HIP_TRY(hipHostFree(bt->pin_out));
        bt->pin_out = nullptr;
        bt->pin_out_bytes = 2 * (8 + nc + nh) + 256;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&bt->pin_out), bt->pin_out_bytes, hipHostMallocDefault));
    }
    HIP_TRY(hipMemcpyAsync(bt->pin_out, bt->error_flag.p, 4, hipMemcpyDeviceToHost, bt->lat_stream));
    if (bt->nq && !bt->results_pinned_now) {
        HIP_TRY(hipMemcpyAsync(bt->pin_out + 8, bt->n_hits.p, nc, hipMemcpyDeviceToHost, bt->lat_stream));
        HIP_TRY(hipMemcpyAsync(bt->pin_out + 8 + nc, bt->hits.p, nh, hipMemcpyDeviceToHost, bt->lat_stream));
    }
    bt->download_enqueued = true;
    return VBM25_OK;
}

int vbm25_batch_device_results(vbm25_batch *bt, void **hits, void **n_hits) {
    if (!bt) return set_error(VBM25_ERR_INVALID, "batch is NULL");
    if (hits) *hits = bt->hits.p;
    if (n_hits) *n_hits = bt->n_hits.p;
    // (whoever reads the records on the device gets them complete after every run: the one-launch form of scan_win_kernel, which
    // leaves a query whose item it gave up to vbm25_batch_fetch, is not used for this batch object any more)
    bt->device_consumer = true;
    return VBM25_OK;
}

static int vbm25_evaluate_batch_impl(vbm25_index *ix, const uint32_t *q_terms, uint32_t n_q, uint32_t n_docs,
                                     const uint64_t *doc_start, const uint32_t *doc_term, const uint32_t *doc_tf,
                                     double *scores) {
    if (!ix || (n_q && !q_terms) || !doc_start || (n_docs && !scores)) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (n_docs == 0) return VBM25_OK;
    const uint64_t n_el = doc_start[n_docs];
    if (n_el && (!doc_term || !doc_tf)) return set_error(VBM25_ERR_INVALID, "NULL argument");
    {  // Query::checked_new, vector.rs:106-110: keys strictly ascending; ids of unknown tokens (>= n_terms) stand
       // wherever their keys sorted and are skipped, so every known id is compared with the last KNOWN one
        bool have = false;
        uint32_t prev = 0;
        for (uint32_t i = 0; i < n_q; ++i) {
            if (q_terms[i] >= ix->n_terms) continue;
            if (have && q_terms[i] <= prev) return set_error(VBM25_ERR_INVALID, "query term ids must be strictly ascending");
            prev = q_terms[i];
            have = true;
        }
    }
    for (uint32_t d = 0; d < n_docs; ++d) {
        if (doc_start[d + 1] < doc_start[d]) return set_error(VBM25_ERR_INVALID, "doc_start not monotone at document %u", d);
        uint32_t prev = 0;
        bool first = true;
        for (uint64_t e = doc_start[d]; e < doc_start[d + 1]; ++e) {
            if (doc_tf[e] == 0) return set_error(VBM25_ERR_INVALID, "document %u: term frequency 0", d);  // Document::checked_new
            if (doc_term[e] >= ix->n_terms) continue;
            if (!first && doc_term[e] <= prev) return set_error(VBM25_ERR_INVALID, "document %u: keys must be strictly ascending", d);
            prev = doc_term[e];
            first = false;
        }
    }
    if (int rc = use_device(ix->device)) return rc;
    if (ix->n_docs == 0) {  // avgdl is 0 / 0 in the reference: NaN scores; report it instead
        return set_error(VBM25_ERR_INVALID, "evaluate on an index without sealed documents");
    }
    DeviceBuffer dq, ds, dt, df, dout;
    int rc = 0;
    if ((rc = dq.upload(q_terms, 4ull * n_q)) || (rc = ds.upload(doc_start, 8ull * (n_docs + 1))) ||
        (rc = dt.upload(doc_term, 4ull * n_el)) || (rc = df.upload(doc_tf, 4ull * n_el)) || (rc = dout.alloc(8ull * n_docs)))
        return rc;
    EvalArgs a{};
    a.n_docs = n_docs;
    a.n_q = n_q;
    a.n_terms = ix->n_terms;
    a.q_terms = dq.as<uint32_t>();
    a.doc_start = ds.as<uint64_t>();
    a.doc_term = dt.as<uint32_t>();
    a.doc_tf = df.as<uint32_t>();
    a.term_idf = ix->term_idf.as<double>();
    a.s1 = ix->dev.s1;
    a.fn_len = ix->fn_len.as<uint32_t>();
    a.k1p1 = ix->k1 + 1.0;
    a.out = dout.as<double>();
    evaluate_kernel<<<(n_docs + 255) / 256, 256>>>(a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(scores, dout.p, 8ull * n_docs, hipMemcpyDeviceToHost));
    return VBM25_OK;
}

int vbm25_evaluate_batch(vbm25_index *ix, const uint32_t *q_terms, uint32_t n_q, uint32_t n_docs,
                         const uint64_t *doc_start, const uint32_t *doc_term, const uint32_t *doc_tf, double *scores) {
    return guarded([&] { return vbm25_evaluate_batch_impl(ix, q_terms, n_q, n_docs, doc_start, doc_term, doc_tf, scores); });
}

// tuning / test aid (not declared in include/vbm25.h): process-wide switches, read when a batch object is created.
// Names: dense_x1000, dense, ne, fused, ne_ratio, dense_items, range_items, range_min_chunk, range_grid, dense_grid, fused_items, arith,
// win, win_force, win_items, win_planes, win_guided, win_grid, win_skew, dbg.
int vbm25_tuning_set(const char *name, long long value) {
    if (!name) return set_error(VBM25_ERR_INVALID, "NULL argument");
    std::lock_guard<std::mutex> guard(g_tune_mutex);
    const std::string n(name);
    if (n == "dense_x1000") g_tune.dense_x1000 = value;
    else if (n == "dense") g_tune.dense = value != 0;
    else if (n == "ne") g_tune.ne = value != 0;
    else if (n == "fused") g_tune.fused = value != 0;
    else if (n == "ne_ratio") g_tune.ne_ratio = (uint32_t)std::max(1ll, value);
    else if (n == "dense_items") g_tune.dense_items = (uint32_t)std::max(256ll, value);
    else if (n == "range_items") g_tune.range_items = (uint32_t)std::max(256ll, value);
    else if (n == "range_min_chunk") g_tune.range_min_chunk = (uint32_t)std::max(128ll, value);
    else if (n == "range_grid") g_tune.range_grid = (uint32_t)std::max(1ll, value);
    else if (n == "dense_grid") g_tune.dense_grid = (uint32_t)std::max(1ll, value);
#ifdef VBM25_DEV
    else if (n == "dbg") g_tune.dbg = (uint32_t)value;  // (scan_win_kernel's timing experiments: development build only)
#endif
    else if (n == "fused_items") g_tune.fused_items = (uint32_t)std::max(0ll, value);
    else if (n == "arith") g_tune.arith = value != 0;
    else if (n == "win") g_tune.win = value != 0;
    else if (n == "win_force") g_tune.win_force = value != 0;
    else if (n == "win_items") g_tune.win_items = (uint32_t)std::max(0ll, value);
    else if (n == "win_planes") g_tune.win_planes = value != 0;
    else if (n == "win_guided") g_tune.win_guided = value != 0;
    else if (n == "rel16_plane") g_tune.rel16_plane = value != 0;
    else if (n == "id16_plane") g_tune.id16_plane = value != 0;
    else if (n == "win_order_arith") g_tune.win_order_arith = value != 0;
    else if (n == "win_grid") g_tune.win_grid = (uint32_t)std::max(0ll, value);
    else if (n == "win_skew") g_tune.win_skew = value != 0;
    else if (n == "win_cut1") g_tune.win_cut1 = int(std::min<long long>(std::max<long long>(value, 1), 998));
    else if (n == "win_cut2") g_tune.win_cut2 = int(std::min<long long>(std::max<long long>(value, 2), 999));
    else if (n == "win_fuse") g_tune.win_fuse = value != 0;
    else return set_error(VBM25_ERR_INVALID, "unknown tuning switch %s", name);
    ++g_tune.generation;
    return VBM25_OK;
}
void vbm25_tuning_reset(void) {
    std::lock_guard<std::mutex> guard(g_tune_mutex);
    const uint32_t gen = g_tune.generation + 1u;
    g_tune = Tuning();
    g_tune.generation = gen;
}

// tuning / test aid (not declared in include/vbm25.h): work items of the last run and how many of them the
// first-choice kernel handed to scan_many_kernel
int vbm25_batch_debug_counts(vbm25_batch *bt, uint32_t *n_items, uint32_t *n_failed) {
    if (!bt || !n_items || !n_failed) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (int rc = use_device(bt->index->device)) return rc;
    HIP_TRY(hipStreamSynchronize(bt->last_stream));
    if ((bt->arith_g || bt->win_g) && bt->state_clean) {  // merge_kernel has cleaned the flags and kept the counts per query
        *n_items = bt->nq * (bt->win_g ? bt->win_g : bt->arith_g);
        std::vector<uint32_t> qf(bt->nq);
        if (bt->nq) HIP_TRY(hipMemcpy(qf.data(), bt->q_failed.p, 4ull * bt->nq, hipMemcpyDeviceToHost));
        *n_failed = 0;
        for (uint32_t x : qf) *n_failed += x;
        return VBM25_OK;
    }
    HIP_TRY(hipMemcpy(n_items, bt->n_items.p, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> f(*n_items);
    if (*n_items) HIP_TRY(hipMemcpy(f.data(), bt->item_failed.p, 4ull * *n_items, hipMemcpyDeviceToHost));
    *n_failed = 0;
    for (uint32_t x : f) *n_failed += x != 0;
    return VBM25_OK;
}

// test aid (not declared in include/vbm25.h): the route the current queries take -- 0 general (plan_kernel), 1 one launch,
// 2 plan-free scan_range_kernel, 3 scan_win_kernel, 4 exhaustive k > 1024
int vbm25_batch_debug_route(vbm25_batch *bt) {
    if (!bt) return -1;
    return bt->bigk ? 4 : bt->fused_g ? 1 : bt->win_g ? 3 : bt->arith_g ? 2 : 0;
}

// test / bench aid (not declared in include/vbm25.h): which kernel the current queries' work goes to.  out[0..2] = queries the host
// classed sparse (scan_win_kernel / scan_range_kernel), dense (scan_dense_kernel: postings >= 0.1 n_docs), many-term (> 16 indexed
// terms: scan_many_kernel); out[3..5] = work items of the last run on the general route by the same classes (0 on the routes whose
// kernels make their items themselves: all of them are the sparse kernel's)
int vbm25_batch_debug_routes(vbm25_batch *bt, uint32_t *out6) {
    if (!bt || !out6) return set_error(VBM25_ERR_INVALID, "NULL argument");
    for (int i = 0; i < 6; ++i) out6[i] = 0;
    if (bt->bigk) return VBM25_OK;
    if (int rc = use_device(bt->index->device)) return rc;
    HIP_TRY(hipStreamSynchronize(bt->last_stream));
    for (uint32_t q = 0; q < bt->nq; ++q) out6[bt->h_dense[q] ? 1 : 0]++;
    if (!bt->fused_g && !bt->win_g && !bt->arith_g) {
        uint32_t n = 0;
        HIP_TRY(hipMemcpy(&n, bt->n_items.p, 4, hipMemcpyDeviceToHost));
        n = std::min(n, bt->max_items);
        std::vector<Item> items(n);
        if (n) HIP_TRY(hipMemcpy(items.data(), bt->items.p, sizeof(Item) * size_t(n), hipMemcpyDeviceToHost));
        for (const Item &it : items) {
            const uint32_t m = it.m & ~ITEM_DENSE;
            out6[3 + (m > 16u ? 2 : (it.m & ITEM_DENSE) ? 1 : 0)]++;
        }
    }
    return VBM25_OK;
}

// test aid (not declared in include/vbm25.h): launches of the last run's scan -- 1: scan_win_kernel merged the queries' lists itself
// (win_fuse), 3: scan_win_kernel, scan_many_kernel, merge_kernel (also after a one-launch run that marked a query), 0: another route
int vbm25_batch_debug_win_launches(vbm25_batch *bt) {
    if (!bt || !bt->win_g) return 0;
    return bt->win_fused_run ? 1 : 3;
}

// -DVBM25_CHECK builds (not declared in include/vbm25.h): the first violated assertion of the scan kernels, then reset.
// out[0] = check code (0: none), out[1] = offending value, out[2] = work item, out[3] = thread
int vbm25_batch_debug_check(vbm25_batch *bt, uint32_t *out4) {  // out4: 16 words
    if (!bt || !out4) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (int rc = use_device(bt->index->device)) return rc;
    HIP_TRY(hipStreamSynchronize(bt->last_stream));
    if (!bt->dbg.p) {
        std::memset(out4, 0, 64);
        return VBM25_OK;
    }
    HIP_TRY(hipMemcpy(out4, bt->dbg.p, 64, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(bt->dbg.p, 0, 64));
    return VBM25_OK;
}

// test aid (not declared in include/vbm25.h): the per-query thresholds the last run ended with (bits of a lower bound of
// each query's k-th best score; the merge drops list entries below them)
int vbm25_batch_debug_theta(vbm25_batch *bt, unsigned long long *out) {
    if (!bt || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (int rc = use_device(bt->index->device)) return rc;
    HIP_TRY(hipStreamSynchronize(bt->last_stream));
    if (bt->nq && bt->theta.p)
        HIP_TRY(hipMemcpy(out, (bt->arith_g || bt->win_g) && bt->state_clean ? bt->theta_last.p : bt->theta.p, 8ull * bt->nq, hipMemcpyDeviceToHost));
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
// profiling builds only (not declared in include/vbm25.h): copy out the phase counters
int vbm25_batch_profile(vbm25_batch *bt, unsigned long long *out, uint32_t n_workgroups) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, bt->prof.p, 8ull * 16 * RNW * std::min<uint32_t>(n_workgroups, R_GRID), hipMemcpyDeviceToHost));
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
    if (!bt || bt->k != k || bt->max_queries < nq || bt->max_terms < n_terms || bt->tune.generation != tuning_snapshot().generation) {
        if (bt) vbm25_batch_destroy(bt);
        ix->scratch = nullptr;
        if (int rc = vbm25_batch_create(ix, std::max(nq, 16u), std::max(n_terms, 256u), k, &bt)) return rc;
        bt->pinned_results = true;  // (host buffers in, host buffers out: nobody reads this batch's records on the device)
        ix->scratch = bt;
    }
    const bool fast = !bt->bigk;
    int rc = vbm25_batch_set_queries_impl(bt, term_ids, q_off, nq, fast);
    if (!rc) rc = vbm25_batch_run_impl(bt, fast ? bt->lat_stream : nullptr);
    if (!rc) rc = vbm25_batch_fetch_impl(bt, hits, n_hits, fast);
    if (rc == VBM25_RETRY_GENERAL) {  // the one-launch route met an item it cannot finish: general route
        bt->fused_g = 0;
        rc = upload_staged(bt);
        if (!rc) rc = vbm25_batch_run_impl(bt, bt->lat_stream);
        if (!rc) rc = vbm25_batch_fetch_impl(bt, hits, n_hits, fast);
    }
    return rc;
}

int vbm25_index_create(const vbm25_index_desc *d, int device, vbm25_index **out) {
    return guarded([&] { return vbm25_index_create_impl(d, device, out); });
}
int vbm25_index_create_from_device(const vbm25_device_segment *ds, vbm25_index **out) {
    return guarded([&] { return vbm25_index_create_from_device_impl(ds, out); });
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

// the shard's part of vbm25_multi_batch_fetch: wait for the part's stream, copy out; an item the one-launch route gave up is
// redone on the general route, as in vbm25_search_batch
static int vbm25_batch_finish_download(vbm25_batch *bt, vbm25_hit *hits, uint32_t *n_hits) {
    const bool fast = !bt->bigk && bt->lat_stream;
    int rc = vbm25_batch_fetch_impl(bt, hits, n_hits, fast);
    if (rc == VBM25_RETRY_GENERAL) {
        bt->fused_g = 0;
        rc = upload_staged(bt);
        if (!rc) rc = vbm25_batch_run_impl(bt, bt->lat_stream);
        if (!rc) rc = vbm25_batch_fetch_impl(bt, hits, n_hits, fast);
    }
    return rc;
}

// ---------------------------------------------------------------------------
// The pipelined boundary (include/vbm25.h): a ring of batch objects, each with its own stream and pinned staging.  submit =
// set_queries (staged) + run, enqueued on the slot's stream; merge_kernel writes the counts and the 24-byte records straight into
// the slot's pinned output buffer (pinned_results: posted writes over PCIe -- no download command on the step, only the 4-byte
// flag is copied); collect = the oldest slot's stream synchronisation and one copy out of its pinned buffer.  The kernels of
// neighbouring batches are launched into each other's tails: on C3 a pipelined step is SHORTER than a step of a loop over resident
// batches on one stream (0.21 against 0.235 ms, profiles/r5_stream_host_time.txt).  With the records downloaded by copy
// commands (three per batch) the same ring took 0.25 ms.
// ---------------------------------------------------------------------------
}  // extern "C"
struct vbm25_stream {
    std::vector<vbm25_batch *> slots;
    uint32_t head = 0, in_flight = 0;  // the oldest batch in flight, their number
    ~vbm25_stream() {
        for (vbm25_batch *b : slots) vbm25_batch_destroy(b);
    }
};
static int stream_create_impl(vbm25_index *ix, uint32_t depth, uint32_t max_queries, uint32_t max_total_terms, uint32_t k, vbm25_stream **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!ix) return set_error(VBM25_ERR_INVALID, "index is NULL");
    if (depth == 0 || depth > 16) return set_error(VBM25_ERR_INVALID, "depth must be 1..16");
    auto s = std::make_unique<vbm25_stream>();
    for (uint32_t i = 0; i < depth; ++i) {
        vbm25_batch *b = nullptr;
        if (int rc = vbm25_batch_create(ix, max_queries, std::max(max_total_terms, 1u), k, &b)) return rc;
        b->pinned_results = true;
        s->slots.push_back(b);
    }
    *out = s.release();
    return VBM25_OK;
}
static int stream_submit_impl(vbm25_stream *s, const uint32_t *term_ids, const uint32_t *q_off, uint32_t nq) {
    if (!s || !q_off) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (s->in_flight == s->slots.size()) return set_error(VBM25_ERR_INVALID, "%u batches in flight: collect one first", s->in_flight);
    vbm25_batch *b = s->slots[(s->head + s->in_flight) % s->slots.size()];
    const bool fast = !b->bigk;
    if (int rc = vbm25_batch_set_queries_impl(b, term_ids, q_off, nq, fast)) return rc;
    if (int rc = vbm25_batch_run_impl(b, fast ? b->lat_stream : nullptr)) return rc;
    if (int rc = vbm25_batch_enqueue_download(b)) return rc;
    ++s->in_flight;
    return VBM25_OK;
}
static int stream_collect_impl(vbm25_stream *s, vbm25_hit *hits, uint32_t *n_hits, uint32_t *nq_out) {
    if (!s) return set_error(VBM25_ERR_INVALID, "stream is NULL");
    if (!s->in_flight) return set_error(VBM25_ERR_INVALID, "no batch in flight");
    vbm25_batch *b = s->slots[s->head];
    if ((!hits || !n_hits) && b->nq) return set_error(VBM25_ERR_INVALID, "NULL argument");
    const int rc = vbm25_batch_finish_download(b, hits, n_hits);
    if (nq_out) *nq_out = b->nq;
    s->head = (s->head + 1) % uint32_t(s->slots.size());
    --s->in_flight;
    return rc;
}
extern "C" {
int vbm25_stream_create(vbm25_index *ix, uint32_t depth, uint32_t max_queries, uint32_t max_total_terms, uint32_t k, vbm25_stream **out) {
    return guarded([&] { return stream_create_impl(ix, depth, max_queries, max_total_terms, k, out); });
}
void vbm25_stream_destroy(vbm25_stream *s) { delete s; }
int vbm25_stream_submit(vbm25_stream *s, const uint32_t *term_ids, const uint32_t *q_off, uint32_t nq) {
    return guarded([&] { return stream_submit_impl(s, term_ids, q_off, nq); });
}
int vbm25_stream_collect(vbm25_stream *s, vbm25_hit *hits, uint32_t *n_hits, uint32_t *nq_out) {
    return guarded([&] { return stream_collect_impl(s, hits, n_hits, nq_out); });
}
int vbm25_stream_in_flight(const vbm25_stream *s) { return s ? int(s->in_flight) : 0; }


// ---------------------------------------------------------------------------
// Several GPUs of one node behind the C ABI (BASELINE.json configs[3]).  The path shards by
// independent queries: the index is replicated, a batch is cut into contiguous shards, nothing is exchanged between the
// GPUs but the replicas themselves -- made ONCE, GPU to GPU (hipMemcpyPeerAsync over xGMI: one host upload, n - 1 peer
// copies, derived arrays included).  The hit records go straight from every GPU to the caller's host buffer (pinned
// staging, one asynchronous copy per device and run): the caller is the host, so a device-side gather (peer copies or
// RCCL) would only add a hop.  One host thread drives all devices: every device has its own batch objects and stream,
// the shards run concurrently, fetch waits for all of them.
// ---------------------------------------------------------------------------
}  // extern "C"

// One host thread per device (round 5): a shard's set_queries -- validation, routing, staging in pinned memory: 80 us per 1024
// queries -- and the enqueue of its scan were done for all the devices by the caller's thread, one after the other; at eight devices
// that was 0.78 ms of host time per step against 0.28 ms of device time (profiles/r5_multi_enqueue.txt).  The workers live as long
// as the vbm25_multi; a call hands every worker its part and waits for all of them.
struct MultiWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;  // (empty: none)
    bool quit = false, done = true;
    int rc = 0;
    char err[sizeof g_error] = "";
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return quit || job; });
            if (quit) return;
            std::function<int()> f = std::move(job);
            job = nullptr;
            lk.unlock();
            int r;
            try {  // (an exception on a worker thread would be std::terminate: it becomes the part's error code)
                r = f();
            } catch (const std::bad_alloc &) {
                r = set_error(VBM25_ERR_NOMEM, "out of host memory in a device worker");
            } catch (const std::exception &e) {
                r = set_error(VBM25_ERR_DEVICE, "exception in a device worker: %s", e.what());
            }
            lk.lock();
            rc = r;
            if (r) std::memcpy(err, g_error, sizeof err);
            done = true;
            cv.notify_all();
        }
    }
};
struct vbm25_multi {
    std::vector<vbm25_index *> replicas;  // [0] is the one uploaded from the host
    vbm25_multi_batch *scratch = nullptr;  // batch object re-used by vbm25_multi_search_batch
    std::vector<std::unique_ptr<MultiWorker>> workers;  // one per replica beyond the first (the caller's thread takes part 0)
    ~vbm25_multi() {
        for (auto &w : workers) {
            {
                std::lock_guard<std::mutex> g(w->m);
                w->quit = true;
            }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
        for (vbm25_index *ix : replicas) vbm25_index_destroy(ix);
    }
    // fn(i) for every part i, concurrently; the first error (by part number) is the call's
    // The workers and their one job slot belong to the vbm25_multi: calls on different vbm25_multi_batch objects of one vbm25_multi from
    // several threads are serialised here (vbm25.h says so).
    std::mutex call_m;
    int each_part(size_t n, const std::function<int(size_t)> &fn) {
        std::lock_guard<std::mutex> call_guard(call_m);
        while (workers.size() + 1 < n) {
            workers.emplace_back(new MultiWorker);
            MultiWorker *w = workers.back().get();
            w->th = std::thread([w] { w->loop(); });
        }
        for (size_t i = 1; i < n; ++i) {
            MultiWorker *w = workers[i - 1].get();
            std::lock_guard<std::mutex> g(w->m);
            w->done = false;
            w->job = [&fn, i] { return fn(i); };
            w->cv.notify_all();
        }
        int rc = 0;
        try {  // (the workers hold a reference to fn: they are always waited for, whatever part 0 does)
            rc = n ? fn(0) : 0;
        } catch (const std::bad_alloc &) {
            rc = set_error(VBM25_ERR_NOMEM, "out of host memory");
        } catch (const std::exception &e) {
            rc = set_error(VBM25_ERR_DEVICE, "exception in part 0: %s", e.what());
        }
        char err[sizeof g_error];
        std::memcpy(err, g_error, sizeof err);
        for (size_t i = 1; i < n; ++i) {
            MultiWorker *w = workers[i - 1].get();
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->done; });
            if (w->rc && !rc) {
                rc = w->rc;
                std::memcpy(err, w->err, sizeof err);
            }
        }
        if (rc) std::memcpy(g_error, err, sizeof err);
        return rc;
    }
};
struct vbm25_multi_batch {
    vbm25_multi *multi = nullptr;
    uint32_t max_queries = 0, max_terms = 0, k = 0, nq = 0;
    std::vector<vbm25_batch *> parts;      // one per replica
    std::vector<uint32_t> lo;              // shard bounds: replica i has the queries [lo[i], lo[i + 1])
    std::vector<std::vector<uint32_t>> off_parts;  // per part: its queries' offsets rebased to 0
    uint32_t tune_generation = 0;          // of the tuning switches its parts copied (vbm25_multi_search_batch rebuilds a stale one)
    ~vbm25_multi_batch() {
        for (vbm25_batch *b : parts) vbm25_batch_destroy(b);
    }
};

namespace {

int clone_index(const vbm25_index *src, int device, vbm25_index **out) {
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return set_error(VBM25_ERR_INVALID, "device %d out of range (%d devices)", device, n_dev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!std::strstr(prop.gcnArchName, "gfx950"))
        return set_error(VBM25_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    if (int rc = use_device(device)) return rc;
    if (device != src->device) {  // direct xGMI copies where the runtime allows them (staged through the host otherwise)
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
        (void)hipGetLastError();
    }
    auto ix = std::make_unique<vbm25_index>();
    ix->device = device;
    ix->n_docs = src->n_docs;
    ix->n_terms = src->n_terms;
    ix->n_blocks = src->n_blocks;
    ix->term_key = src->term_key;
    ix->term_df_host = src->term_df_host;
    ix->term_win_host = src->term_win_host;
    ix->n_win = src->n_win;
    ix->k1 = src->k1;
    ix->device_bytes = src->device_bytes;
    for (auto member : INDEX_BUFFERS) {
        const DeviceBuffer &from = src->*member;
        if (!from.p) continue;
        DeviceBuffer &to = (*ix).*member;
        if (int rc = to.alloc(from.bytes)) return rc;
        if (from.bytes) HIP_TRY(hipMemcpyPeerAsync(to.p, device, from.p, src->device, from.bytes, nullptr));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    fill_dev(ix.get());
    ix->dev.blob_bytes = src->dev.blob_bytes;
    ix->dev.blk_ub_attained = src->dev.blk_ub_attained;
    *out = ix.release();
    return VBM25_OK;
}

int multi_create_impl(const vbm25_index_desc *desc, const int *devices, int n_devices, vbm25_multi **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices <= 0) return set_error(VBM25_ERR_INVALID, "no devices given");
    auto m = std::make_unique<vbm25_multi>();
    vbm25_index *first = nullptr;
    if (int rc = vbm25_index_create(desc, devices[0], &first)) return rc;
    m->replicas.push_back(first);
    for (int i = 1; i < n_devices; ++i) {
        vbm25_index *r = nullptr;
        if (int rc = clone_index(first, devices[i], &r)) return rc;
        m->replicas.push_back(r);
    }
    *out = m.release();
    return VBM25_OK;
}

int multi_batch_create_impl(vbm25_multi *m, uint32_t max_queries, uint32_t max_total_terms, uint32_t k, vbm25_multi_batch **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!m) return set_error(VBM25_ERR_INVALID, "multi is NULL");
    if (!max_queries) return set_error(VBM25_ERR_INVALID, "max_queries is 0");
    auto mb = std::make_unique<vbm25_multi_batch>();
    mb->multi = m;
    mb->max_queries = max_queries;
    mb->max_terms = max_total_terms;
    mb->k = k;
    const uint32_t n = uint32_t(m->replicas.size());
    const uint32_t per = (max_queries + n - 1) / n;  // a shard is at most this many queries; it may hold all the terms
    for (vbm25_index *ix : m->replicas) {
        vbm25_batch *b = nullptr;
        if (int rc = vbm25_batch_create(ix, per, std::max(max_total_terms, 1u), k, &b)) return rc;
        b->pinned_results = true;
        mb->parts.push_back(b);
    }
    mb->lo.assign(n + 1, 0);
    mb->tune_generation = tuning_snapshot().generation;
    *out = mb.release();
    return VBM25_OK;
}

int multi_batch_set_queries_impl(vbm25_multi_batch *mb, const uint32_t *term_ids, const uint32_t *q_off, uint32_t nq) {
    if (!mb || !q_off) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (nq > mb->max_queries) return set_error(VBM25_ERR_INVALID, "%u queries exceed the batch capacity %u", nq, mb->max_queries);
    if (q_off[0] != 0) return set_error(VBM25_ERR_INVALID, "q_off[0] must be 0");
    const uint32_t n = uint32_t(mb->parts.size());
    mb->nq = nq;
    for (uint32_t i = 0; i <= n; ++i) {  // contiguous, balanced shards (the first nq % n get one query more)
        const uint32_t base = nq / n, rem = nq % n;
        mb->lo[i] = i * base + std::min(i, rem);
    }
    for (uint32_t i = 0; i < n; ++i)
        if (q_off[mb->lo[i + 1]] < q_off[mb->lo[i]]) return set_error(VBM25_ERR_INVALID, "q_off not monotone");
    mb->off_parts.resize(n);
    // every shard by its device's own host thread: validation, routing and staging of the shards run side by side
    return mb->multi->each_part(n, [&](size_t i) -> int {
        const uint32_t a = mb->lo[i], b = mb->lo[i + 1];
        std::vector<uint32_t> &off = mb->off_parts[i];
        off.resize(size_t(b - a) + 1);
        for (uint32_t q = a; q <= b; ++q) off[q - a] = q_off[q] - q_off[a];
        // staged in the part's pinned memory, copied on its own stream: the devices' uploads overlap
        return vbm25_batch_set_queries_impl(mb->parts[i], term_ids ? term_ids + q_off[a] : nullptr, off.data(), b - a, true);
    });
}

int multi_batch_run_impl(vbm25_multi_batch *mb) {
    if (!mb) return set_error(VBM25_ERR_INVALID, "batch is NULL");
    // every device on its own stream, enqueued by its own host thread: the shards run concurrently
    return mb->multi->each_part(mb->parts.size(), [&](size_t i) -> int {
        vbm25_batch *b = mb->parts[i];
        if (!b->nq) return VBM25_OK;
        if (int rc = vbm25_batch_run_impl(b, b->bigk ? nullptr : b->lat_stream)) return rc;
        return vbm25_batch_enqueue_download(b);  // the shard's records to pinned host memory, behind its scan
    });
}

int multi_batch_fetch_impl(vbm25_multi_batch *mb, vbm25_hit *hits, uint32_t *n_hits) {
    if (!mb || (!hits && mb->nq) || (!n_hits && mb->nq)) return set_error(VBM25_ERR_INVALID, "NULL argument");
    // every part is drained -- by its device's own host thread -- whatever another one returns: no stream is left with a download
    // pending that the next fetch would take for its own; the first error (by part number) is the call's
    return mb->multi->each_part(mb->parts.size(), [&](size_t i) -> int {
        vbm25_batch *b = mb->parts[i];
        if (!b->nq) return VBM25_OK;
        const int rc = vbm25_batch_finish_download(b, hits + size_t(mb->lo[i]) * mb->k, n_hits + mb->lo[i]);
        b->download_enqueued = false;
        return rc;
    });
}

}  // namespace

extern "C" {

int vbm25_multi_create(const vbm25_index_desc *desc, const int *devices, int n_devices, vbm25_multi **out) {
    return guarded([&] { return multi_create_impl(desc, devices, n_devices, out); });
}
void vbm25_multi_destroy(vbm25_multi *m) {
    if (!m) return;
    if (m->scratch) vbm25_multi_batch_destroy(m->scratch);
    delete m;
}
int vbm25_multi_device_count(const vbm25_multi *m) { return m ? int(m->replicas.size()) : 0; }
int vbm25_multi_index(vbm25_multi *m, int i, vbm25_index **out) {
    if (!m || !out || i < 0 || size_t(i) >= m->replicas.size()) return set_error(VBM25_ERR_INVALID, "bad argument");
    *out = m->replicas[size_t(i)];
    return VBM25_OK;
}
int vbm25_multi_batch_create(vbm25_multi *m, uint32_t max_queries, uint32_t max_total_terms, uint32_t k, vbm25_multi_batch **out) {
    return guarded([&] { return multi_batch_create_impl(m, max_queries, max_total_terms, k, out); });
}
void vbm25_multi_batch_destroy(vbm25_multi_batch *mb) { delete mb; }
int vbm25_multi_batch_set_queries(vbm25_multi_batch *mb, const uint32_t *term_ids, const uint32_t *q_off, uint32_t nq) {
    return guarded([&] { return multi_batch_set_queries_impl(mb, term_ids, q_off, nq); });
}
int vbm25_multi_batch_run(vbm25_multi_batch *mb) {
    return guarded([&] { return multi_batch_run_impl(mb); });
}
int vbm25_multi_batch_fetch(vbm25_multi_batch *mb, vbm25_hit *hits, uint32_t *n_hits) {
    return guarded([&] { return multi_batch_fetch_impl(mb, hits, n_hits); });
}
int vbm25_multi_search_batch(vbm25_multi *m, const uint32_t *term_ids, const uint32_t *q_off, uint32_t nq, uint32_t k,
                             vbm25_hit *hits, uint32_t *n_hits) {
    return guarded([&]() -> int {
        if (!m || !q_off) return set_error(VBM25_ERR_INVALID, "NULL argument");
        if (nq == 0) return k ? VBM25_OK : set_error(VBM25_ERR_INVALID, "number of needed rows is set to 0");
        vbm25_multi_batch *mb = m->scratch;
        const uint32_t n_terms = q_off[nq] ? q_off[nq] : 1;
        if (!mb || mb->k != k || mb->max_queries < nq || mb->max_terms < n_terms || mb->tune_generation != tuning_snapshot().generation) {
            if (mb) vbm25_multi_batch_destroy(mb);
            m->scratch = nullptr;
            if (int rc = multi_batch_create_impl(m, std::max(nq, 16u), std::max(n_terms, 256u), k, &mb)) return rc;
            m->scratch = mb;
        }
        if (int rc = multi_batch_set_queries_impl(mb, term_ids, q_off, nq)) return rc;
        if (int rc = multi_batch_run_impl(mb)) return rc;
        return multi_batch_fetch_impl(mb, hits, n_hits);
    });
}

}  // extern "C"
