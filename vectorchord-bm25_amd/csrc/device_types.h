// device_types.h -- device-side views of the index and of one batch; constants of the kernels.
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, block_fetch, topk_reg, scan_range, scan_dense, scan_many, merge (scan_win.hip:
// device_types, decode, topk_lds, block_fetch, topk_reg, scan_win).

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
    const double *term_kth_ub;   // derived: 9 per term -- the 2^i-th largest block maximum of the term (the scores themselves, no margin;
                                 // 0 when the term has fewer blocks), NULL when the block maxima are not attained
    const uint8_t *blob;
    const uint8_t *post_fn;      // derived: fieldnorm of every posting, 128 bytes per block
    const uint32_t *post_rel16;  // derived: 64 words per block -- word l = (id[2l + 1] - min_doc) << 16 | (id[2l] - min_doc) of a
                                 // full bit-packed block that spans < 2^16 documents (rel16_block), undefined for the others
    const uint32_t *post_tfn;    // derived: 64 words per block -- word l = tf[2l] | tf[2l + 1] << 8 | fieldnorm[2l] << 16 |
                                 // fieldnorm[2l + 1] << 24 of a full block whose term frequencies are bit-packed in <= 7 bits
                                 // (tfn_block), undefined for the others
    const uint16_t *doc_payload;
    const double *s1;            // 256 entries
    // scan_win_kernel's planes (derived; the window-major view of the same postings):
    const uint32_t *post_id16;   // 64 words per block: word l = (id[2l + 1] & 0xffff) << 16 | (id[2l] & 0xffff) -- the low 16 bits of EVERY
                                 // posting's document id in posting order (tails included): the bit a posting owns in the filter of
                                 // its 2^16-document window.  A term's postings are contiguous (only its last block is short)
    const uint32_t *win_off;     // per qualifying term n_win + 1 entries: entry w = postings of the term with document < w << 16
    const uint32_t *term_win;    // per term: first entry in win_off, NONE32 when the term has no table (too few postings)
    uint32_t n_win;              // windows of 2^16 documents: ceil(n_docs / 65536)
    unsigned long long blob_bytes;  // bytes of blob that hold block bodies (the allocation has slack behind them)
    uint32_t blk_ub_attained;    // 1: every blk_ub is the score of a posting of its block (the index came with block WAND pairs)
};

// Blocks whose ids scan_range_kernel reads from post_rel16 (one coalesced word per lane, two adds) instead of unpacking the
// delta stream: full blocks (128 postings, bit-packed, not raw) that span at most 65535 documents.
__device__ __forceinline__ bool rel16_block(uint32_t min_doc, uint32_t max_doc, uint32_t w) {
    return ((w >> 8) & 0xffu) < 32u && (w & 0xffu) == 128u && max_doc - min_doc <= 0xffffu;
}

// Blocks whose term frequencies and fieldnorm bytes scan_dense_kernel reads from post_tfn: full blocks, tf fields of at most
// 7 bits.
__device__ __forceinline__ bool tfn_block(uint32_t w) { return (w & 0xffu) == 128u && ((w >> 16) & 0xffu) <= 7u; }

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
    uint32_t *item_order;      // plan_kernel: the items longest first -- the order the persistent workgroups draw them in
    uint32_t *item_failed;     // per item: != 0 = the first-choice kernel gave the item up, scan_many_kernel redoes it
    unsigned long long *prof;  // VBM25_PROFILE builds: 16 counters per wave
    uint32_t *work_ctr;        // [0] next item of the range kernel, [1] of the dense kernel (reset by plan_kernel)
    uint32_t *hist;            // per query: CUR_HB score buckets, documents accepted by any item
    uint32_t lpi;              // result lists per item (scan_range_kernel: one per wave; the others use list 0)
    uint32_t range_max_terms;  // scan_range_kernel takes the queries with at most this many terms (0: off)
    uint32_t ne_on;            // MaxScore split in scan_range_kernel (non-essential lists looked up, not scanned)
    uint32_t ne_ratio;         // a non-essential list must be this many times longer than the essential lists together
    uint32_t dense_on;         // dense queries of <= D_T terms take scan_dense_kernel (its items come from work_ctr[1])
    uint32_t fused_g;          // scan_range_kernel alone: items per query made in the kernel, lists merged by the query's last workgroup (0: off)
    uint32_t *fused_state;     // [0] workgroups that left, [1 + q] finished items of query q; zero between launches
    uint32_t merge_marked;     // merge_kernel: only the queries whose n_hits is NONE32
    uint32_t merge_clean;      // merge_kernel leaves the per-launch state (thresholds, histogram, list counts, failure flags, item counters)
                               // zero for the next launch: the route without plan_kernel (fused_g != 0 in the general instantiations)
    uint32_t many_expected;    // scan_many_kernel: 0 = the host knows of no item for it; it looks at fail_any and leaves
    uint32_t order_on;         // the route without plan_kernel: item_order holds the host's longest-first order of the items
    uint32_t *fail_any;        // != 0: a first-choice kernel gave an item up in this launch
    uint32_t *q_failed;        // per query: items the first-choice kernels gave up in the last launch (merge_kernel, merge_clean)
    unsigned long long *theta_last;  // per query: the threshold the last launch ended with (merge_kernel, merge_clean)
    uint32_t max_items;        // capacity of items / item_failed; res_* hold max_items * lpi lists of k entries
    uint32_t win_g;            // scan_win_kernel: items (runs of 2^16-document windows) per query, one result list per item
    uint32_t q_stride;         // != 0: every query of the batch has this many terms -- query q is term_ids[q_stride q ..): nobody loads q_off
    uint32_t win_cut[17];      // ... item `part` of a query = the windows [win_cut[part], win_cut[part + 1]) when win_g <= 16 and
                               // win_cut[win_g] != 0 (runs of decreasing length, handed out longest first: the last items drawn
                               // are the short ones); equal runs n_win part / win_g otherwise
    uint32_t win_fuse;         // scan_win_kernel merges a query's lists itself (the wave that finishes the query's last item) and leaves the
                               // per-launch state zero: no scan_many_kernel, no merge_kernel behind it (a query with an item given up gets n_hits = NONE32)
    uint32_t win_dbg;          // development switch of scan_win_kernel (timing experiments only, wrong results; scan_win.h)
    const uint32_t *id16_fb;   // an index without the post_id16 plane (round 6): per term position of term_ids the first 256-byte block of that
                               // term's low-16-bit ids in the batch's scratch plane, which decode_id16_kernel fills from the blob ahead of
                               // scan_win_kernel (ix.post_id16 then points at the scratch plane); NULL: the index's own plane, at term_first_block
    uint32_t *dbg;             // -DVBM25_CHECK builds: [0] first violated check (0: none), [1] value, [2] item, [3] thread
};

// The arguments of the scan kernels (DevIndex ix, DevBatch bt) as they lie in the kernarg segment.  The pointers that only an
// item's setup, its end and the rare paths need are read from there where they are used (cold_args): kept in SGPRs for the whole
// kernel they push the loops' uniform state into VGPRs and the VGPRs into scratch.
struct KernArgs {
    DevIndex ix;
    DevBatch bt;
};
typedef const __attribute__((address_space(4))) KernArgs *KernArgsP;
__device__ __forceinline__ KernArgsP cold_args() {
    KernArgsP p = (KernArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));  // opaque: the loads stay where they are written
    return p;
}

// Bounds / consistency assertions of the scan kernels, compiled in by -DVBM25_CHECK only (tools/dense_stress.py): the
// first violation is recorded in bt.dbg and read back with vbm25_batch_debug_check.
#ifdef VBM25_CHECK
#define VCHK(cond, code, val)                                                     \
    do {                                                                          \
        if (!(cond) && atomicCAS(&bt.dbg[0], 0u, (uint32_t)(code)) == 0u) {       \
            bt.dbg[1] = (uint32_t)(val);                                          \
            bt.dbg[2] = S.item;                                                   \
            bt.dbg[3] = threadIdx.x;                                              \
        }                                                                         \
    } while (0)
#else
#define VCHK(cond, code, val) do { } while (0)
#endif

constexpr int WG = 256;                // scan_many_kernel's workgroup
constexpr int NW = WG / 64;
constexpr int SLOTS_LOG2 = 12;
constexpr int SLOTS = 1 << SLOTS_LOG2;  // hash table slots / window documents per scan_many_kernel workgroup
constexpr int CAP_BLOCKS = SLOTS / 2 / 128;  // blocks admitted per tile in hash mode
constexpr int MAX_TERMS = 1024;        // indexed terms per query handled on the GPU (scan_many_kernel's LDS arrays)
constexpr uint32_t EMPTY = 0xffffffffu;
constexpr uint32_t TARGET_ITEMS = 1536;  // work items of a batch on the scan_many_kernel route
constexpr uint32_t MIN_CHUNK_POSTINGS = 8192;
constexpr int PLAN_WG = 1024;
constexpr int PLAN_BUCKETS = 512;      // of plan_kernel's sort of the items by length (<= PLAN_WG)
constexpr int CUR_HB = 256;            // score buckets of the per-query histogram of accepted documents
constexpr int REG_K = 256;                // largest k whose running top-k lives in registers
constexpr uint32_t NONE32 = 0xffffffffu;
constexpr int KTH_LEVELS = 9;            // term_kth_ub: the 2^i-th largest block maximum of a term, i = 0..8
