// scan_range.h -- scan_range_kernel: the doc-range formulation (queries of <= RT indexed terms, k <= REG_K).
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25.
//
// One 8-wave workgroup per work item (query x doc range), persistent, items from an atomic counter.  The
// item is cut into TILES: a tile is a doc range [tlo, thi) that holds at most R_NBLK posting blocks of the
// query's terms (every block whose min_doc < thi; per-term quotas proportional to df make thi).  Wave 0 is
// the PLANNER (tile n + 1 while the others work on tile n; it also polls the query's shared threshold),
// waves 1..7 are WORKERS with RB blocks each.  A tile is three phases separated by two LDS-only barriers:
//
//   S1  workers get the two ids of every lane and block from ONE word of post_rel16 (ids relative to the block's
//       first document, a plane derived at index creation; fetched one tile ahead, right after the barrier that
//       publishes the plan; blocks without a plane word -- tails, raw, wide -- are decoded from the blob), stage the
//       ids in LDS (one 128-id row per block) and mark every in-range id in the tile's SEEN filter with ONE 32-bit LDS
//       atomic: word (x >> 5) mod 4096, bit x mod 32, x = id - tlo -- exact when the tile is <= 2^17 documents wide; wider
//       tiles add a second, hashed bit in the same word (a blocked Bloom filter: both bits travel in one
//       atomic, so two postings of a document that race still see each other).  A mark that was already there
//       = a SECOND ARRIVAL: the document may sit in two lists.  The wave collects those and inserts them in
//       one pass into a hash set: one ROW per document.
//   S2  lanes = (row, term): binary search of the row's document in the term's staged blocks; a posting found
//       gets its tf field / fieldnorm byte read, Cache::evaluate (bm25.rs:355-358) -> contrib[row][term], done
//       bit.  Few instructions, long LDS chains: the other workgroup of the CU fills the issue slots.
//   S3  lanes = rows: sum of the row in ascending key order (evaluate.rs:43-72 order; absent terms add 0.0,
//       exact) -> offer.  Then the COLD pass, per worker over its own blocks: a block whose upper bound
//       (search.rs:377-380, evaluated once per index) reaches the threshold has every posting without a done
//       bit scored on its own; all other blocks never have their tf / fieldnorm bytes read.
//
// The vector pipes are busy about half of the launch (PMC + measured issue rates): S1 is
// bound by instruction issue and written for instruction count -- branch-free over the wave's eight blocks,
// 32-bit filter words, uniform values in SGPRs --, S2 / S3 and the two barriers are bound by latency.
//
// Every wave keeps its own top-k in registers (RegTopK); the k-th scores are shared through LDS, the query's
// 64-bit atomicMax word and the 256-bucket histogram.  Lists go to res_* at
// item * lpi + wave; merge_kernel merges them.  A tile whose rows overflow hands the item to
// scan_many_kernel (item_failed).

#ifndef VBM25_RNW
#define VBM25_RNW 8
#define VBM25_RB 8
#define VBM25_RWPS 4
#ifndef VBM25_DECODE_GROUP
#define VBM25_DECODE_GROUP 4  // blocks whose raw words are in flight together when the index has no post_rel16 plane
#endif
#define VBM25_RLIST 128
#endif
constexpr int RNW = VBM25_RNW;       // waves per workgroup: planner + 7 workers (16 waves x 4 blocks measured 13 % slower)
constexpr int RWG = RNW * 64;
constexpr int RB = VBM25_RB;         // blocks per worker per tile
constexpr int R_NBLK = (RNW - 1) * RB;  // blocks per tile (slots = lanes 0..R_NBLK-1 of the planner wave)
static_assert(R_NBLK <= 64, "one planner lane per block of a tile");
static_assert(RB % 2 == 0, "a worker row of the plan is read two entries at a time");
constexpr int R_BM_WORDS = 4096;     // 2^17 bits; word R_BM_WORDS is the trash word of out-of-range postings
constexpr uint32_t R_BM_EXACT = 1u << 17;
constexpr int R_HS_LOG2 = 10;  // > rows + every lane of the workers inserting at once: the table never fills

constexpr int R_HS = 1 << R_HS_LOG2;  // slots of the second-arrival hash set
constexpr int R_ROWS = 192;           // rows (documents with a second arrival) per tile at RT = 8; 96 at RT = 16
constexpr uint32_t R_TARGET_ITEMS = 1024;
constexpr uint32_t R_MIN_CHUNK_POSTINGS = 16384;
constexpr uint32_t R_GRID = 512;      // persistent workgroups: 256 CUs x 2 (KMAX <= 64; 1 per CU above)
constexpr int R_PLAN_RING = 3;
constexpr int R_LIST = VBM25_RLIST;   // second arrivals per wave per tile; more than that: scan_many_kernel
constexpr int R_STAGE_STRIDE = 130;   // words between staged rows: 128 would put the same column of every row on one LDS bank (S2 probes columns)

template <int RT>
struct RangeLds {
    uint32_t bm[R_BM_WORDS + 4];
    uint32_t stage[R_NBLK * R_STAGE_STRIDE];  // one row of 128 ids per block
    uint32_t hkeys[R_HS];
    double contrib[R_ROWS * 8];        // rows x RT
    uint32_t done[R_NBLK * 4];
    uint32_t mdoc[2][R_ROWS];
    uint16_t mslot[2][R_ROWS];
    uint4 pm[R_PLAN_RING][R_NBLK];     // {min_doc, max_doc, off8, n | md << 8 | mt << 16 | wand_fn << 24}
    uint2 pa[R_PLAN_RING][R_NBLK];     // {block index, term}
    // the same plan by worker: row w - 1 = the RB entries (w - 1) + (RNW - 1) i of worker w as {block index, first document}, slots
    // beyond the plan filled with its last block -- a worker reads its row with four 16-byte loads (the prefetch of the
    // plane words and S1 read eight entries one by one before: 12 LDS instructions and their address arithmetic per wave and tile)
    alignas(16) uint2 pw[R_PLAN_RING][RNW - 1][RB];
    uint32_t coldw[R_PLAN_RING][RNW];  // per worker: its entries whose upper bound reaches the threshold
    double pub[R_PLAN_RING][R_NBLK];   // block upper bound (read for the entries marked cold only)
    uint4 hdr[R_PLAN_RING];            // {tlo, thi, blocks, -}
    uint8_t ptb[R_PLAN_RING][RT + 4];  // first plan entry of each term (entries of a term are contiguous)
    double s1[256];
    double t_s0[RT];
    double t_ub[RT];                   // token upper bound x (1 + 1e-12)
    double t_cum[RT + 1];              // sum of the p smallest token upper bounds
    uint8_t t_rank[RT];                // position of the term in ascending upper-bound order
    uint8_t t_ord[RT];                 // ... and the term at a position
    uint8_t t_best[RT + 1];            // largest admissible non-essential prefix <= p
    uint32_t t_b0[RT], t_b1[RT];       // block range of the term
    double hscale;
    unsigned long long theta;          // bits of a lower bound of the query's k-th best score
    uint32_t nmulti[2];
    uint32_t item, q, lo, hi, mq, fail;
    uint32_t scratch[64];
    uint32_t list[RNW][R_LIST];        // per wave: second arrivals of a tile, inserted in one pass
    uint32_t lcnt[RNW];
    // planner state (wave 0), kept here between its turns so that the workers do not carry it in registers:
    // per lane {cur, end, quota, base, slot term, slot offset, df, rank, cur before the last two plans}, then
    // the uniform words {ne, relax, cap, good, tlo, tlo before the last two plans}
    uint32_t pl[10][64];
    uint32_t plu[8];
};

// First block of [b0, b1) whose max_doc >= d (b1 if none): guess by interpolation over the document space,
// gallop, then bisect (Cursor::seek_block, search.rs:412-431, without walking the summaries one by one).
__device__ __forceinline__ uint32_t r_first_block_ge(const DevIndex &ix, uint32_t b0, uint32_t b1, uint32_t d) {
    uint32_t lo_b = b0, hi_b = b1;
    if (d != 0 && b1 > b0) {
        uint32_t g = b0 + (uint32_t)((unsigned long long)(b1 - b0) * d / ix.n_docs);
        if (g >= b1) g = b1 - 1;
        if (ix.blk_max_doc[g] < d) {
            lo_b = g + 1;
            for (uint32_t step = 1; lo_b < hi_b; step *= 4) {
                const uint32_t p = min(lo_b + step - 1, hi_b - 1);
                if (ix.blk_max_doc[p] < d) lo_b = p + 1;
                else {
                    hi_b = p;
                    break;
                }
            }
        } else {
            hi_b = g;
            for (uint32_t step = 1; lo_b < hi_b; step *= 4) {
                const uint32_t p = hi_b - lo_b >= step ? hi_b - step : lo_b;
                if (ix.blk_max_doc[p] >= d) hi_b = p;
                else {
                    lo_b = p + 1;
                    break;
                }
            }
        }
        while (lo_b < hi_b) {
            const uint32_t mid = (lo_b + hi_b) >> 1;
            if (ix.blk_max_doc[mid] < d) lo_b = mid + 1; else hi_b = mid;
        }
    }
    return lo_b;
}

template <int KMAX, int RT, bool FUSED = false>
#ifndef VBM25_RWPS_BIGK
#define VBM25_RWPS_BIGK VBM25_RWPS
#endif
__global__ void __launch_bounds__(RWG, KMAX > 64 ? VBM25_RWPS_BIGK : VBM25_RWPS) scan_range_kernel(DevIndex ix, DevBatch bt) {
    static_assert(KMAX <= REG_K, "register top-k only");
    static_assert(RT == 8 || RT == 16, "row stride");
    constexpr int RK = KMAX / 64;
    constexpr int LRT = RT == 8 ? 3 : 4;
    __shared__ RangeLds<RT> S;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t k = bt.k;
    // bt.fused_g != 0 (a handful of queries through vbm25_search_batch): no plan_kernel and no merge_kernel -- every query is
    // cut into fused_g equal document ranges right here, the last workgroup to finish a query merges its lists into the
    // hits and leaves the per-launch state (threshold, histogram, counters) clean for the next launch.
    // The general instantiation with bt.fused_g != 0 (every query of the batch sparse: vbm25_batch_run without plan_kernel):
    // the same items, made right here; merge_kernel merges and cleans.
    const uint32_t fused_g = bt.fused_g;
    const uint32_t n_items = fused_g ? bt.nq * fused_g : *cold_args()->bt.n_items;
    for (uint32_t i = tid; i < 256; i += RWG) S.s1[i] = ix.s1[i];

#ifdef VBM25_PROFILE
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif
    // planner state (wave 0 only): lane t = term t, lane s = plan slot s
    uint32_t p_cur = 0, p_end = 0, p_quota = 0, p_base = 0, p_st = NONE32, p_so = 0, p_df = 0, p_rank = 0, p_ne = 0, p_relax = 0;
    // blocks a tile may take (halved when a tile overflows its rows, see the retry below) and the planner's state
    // before its last two plans
    uint32_t p_cap = R_NBLK, p_good = 0, sv_cur[2] = {0, 0}, sv_tlo[2] = {0, 0};

    for (;;) {
        __syncthreads();  // previous item fully done with LDS
        if (tid == 0) {
            const KernArgsP cd = cold_args();  // (the arguments only an item's setup and end need are read where they are used)
            const uint32_t drawn = atomicAdd(cd->bt.work_ctr, 1u);
            // plan_kernel's order, or (the route without plan_kernel) the host's: longest first -- the last items drawn decide when the launch ends
            S.item = (!fused_g || (!FUSED && cd->bt.order_on)) && drawn < n_items ? cd->bt.item_order[drawn] : drawn;
        }
        for (uint32_t i = tid; i < R_HS; i += RWG) S.hkeys[i] = EMPTY;
        for (uint32_t i = tid; i < R_BM_WORDS + 4; i += RWG) S.bm[i] = 0;
        for (uint32_t i = tid; i < R_ROWS * 8; i += RWG) S.contrib[i] = 0.0;
        if (tid < R_NBLK * 4) S.done[tid] = 0;
        if (tid < RNW) S.lcnt[tid] = 0;
        __syncthreads();
        const uint32_t item = uni(S.item);
        if (item >= n_items) break;
        Item it;
        if (fused_g) {
            it.q = item / fused_g;
            const uint32_t part = item - it.q * fused_g;
            it.doc_lo = (uint32_t)((unsigned long long)ix.n_docs * part / fused_g);
            it.doc_hi = (uint32_t)((unsigned long long)ix.n_docs * (part + 1) / fused_g);
            it.m = 0;  // the host sends only sparse queries of <= RT indexed terms this way
        } else {
            it = cold_args()->bt.items[item];
        }
        if (it.m > (uint32_t)RT) continue;  // more terms or dense (ITEM_DENSE): the other kernels'
        PROF_T(t_item);
        const uint32_t q = uni(it.q), lo = uni(it.doc_lo), hi = uni(it.doc_hi);
        uint32_t *hrow = bt.hist + (size_t)q * CUR_HB;

        // ---- threshold poll (wave 0): the query's published k-th score and the histogram of accepted documents
        unsigned long long pg = 0;
        uint32_t pc[4] = {0, 0, 0, 0};
        auto poll_request = [&]() {
            pg = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 4; ++i) pc[i] = __hip_atomic_load(&hrow[4 * lane + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto poll_consume = [&]() {
            unsigned long long th = ((unsigned long long)uni((uint32_t)(pg >> 32)) << 32) | uni((uint32_t)pg);
            const uint32_t own = pc[0] + pc[1] + pc[2] + pc[3];
            const uint32_t incl = wave_incl_scan_u32(own);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t above = total - incl;  // documents in the buckets of higher lanes
            const unsigned long long hit = __ballot(above + own >= k);
            if (hit) {
                const uint32_t hl = 63u - (uint32_t)__builtin_clzll(hit);
                uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)above, (int)hl), b = 4 * hl;
                const uint32_t c3 = (uint32_t)__builtin_amdgcn_readlane((int)pc[3], (int)hl);
                const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)pc[2], (int)hl);
                const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)pc[1], (int)hl);
                if (a + c3 >= k) b += 3;
                else if (a + c3 + c2 >= k) b += 2;
                else if (a + c3 + c2 + c1 >= k) b += 1;
                // a score lands in bucket b only if score * hscale >= b (up to one rounding)
                const double edge = ((double)b / S.hscale) * (1.0 - 1e-12);
                const unsigned long long eb2 = (unsigned long long)__double_as_longlong(edge);
                if (eb2 > th) th = eb2;
            }
            if (lane == 0) atomicMax(&S.theta, th);
        };

        auto theta_lds = [&]() -> unsigned long long {
            unsigned long long th = S.theta;
            return ((unsigned long long)uni((uint32_t)(th >> 32)) << 32) | uni((uint32_t)th);
        };
        // ---- tile planner (wave 0)
        uint32_t p_tlo = lo;
        auto pl_load = [&]() {
            p_cur = S.pl[0][lane];
            p_end = S.pl[1][lane];
            p_quota = S.pl[2][lane];
            p_base = S.pl[3][lane];
            p_st = S.pl[4][lane];
            p_so = S.pl[5][lane];
            p_df = S.pl[6][lane];
            p_rank = S.pl[7][lane];
            sv_cur[0] = S.pl[8][lane];
            sv_cur[1] = S.pl[9][lane];
            p_ne = uni(S.plu[0]);
            p_relax = uni(S.plu[1]);
            p_cap = uni(S.plu[2]);
            p_good = uni(S.plu[3]);
            p_tlo = uni(S.plu[4]);
            sv_tlo[0] = uni(S.plu[5]);
            sv_tlo[1] = uni(S.plu[6]);
        };
        auto pl_store = [&]() {
            S.pl[0][lane] = p_cur;
            S.pl[1][lane] = p_end;
            S.pl[2][lane] = p_quota;
            S.pl[3][lane] = p_base;
            S.pl[4][lane] = p_st;
            S.pl[5][lane] = p_so;
            S.pl[6][lane] = p_df;
            S.pl[7][lane] = p_rank;
            S.pl[8][lane] = sv_cur[0];
            S.pl[9][lane] = sv_cur[1];
            if (lane == 0) {
                S.plu[0] = p_ne;
                S.plu[1] = p_relax;
                S.plu[2] = p_cap;
                S.plu[3] = p_good;
                S.plu[4] = p_tlo;
                S.plu[5] = sv_tlo[0];
                S.plu[6] = sv_tlo[1];
            }
        };
        // quotas of the essential terms: the 64 candidate slots (one per planner lane) shared in proportion to df,
        // at least one each; slot -> (term, offset)
        auto assign_quotas = [&]() {
            const uint32_t m = uni(S.mq);
            const bool ess = lane < m && p_rank >= p_ne;
            unsigned long long sumdf = 0;
            uint32_t ne = 0;
            for (uint32_t t = 0; t < m; ++t) {
                const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)p_rank, (int)t);
                if (rk >= p_ne) {
                    sumdf += (uint32_t)__builtin_amdgcn_readlane((int)p_df, (int)t);
                    ++ne;
                }
            }
            p_quota = 0;
            if (ess) {
                p_quota = (uint32_t)(((unsigned long long)(64 - ne) * p_df) / sumdf);
                if (p_quota < 1) p_quota = 1;
            }
            const uint32_t incl = wave_incl_scan_u32(p_quota);
            p_base = incl - p_quota;  // lanes >= m: total
            p_st = NONE32;
            p_so = 0;
            for (uint32_t t = 0; t < m; ++t) {
                const uint32_t bt0 = (uint32_t)__builtin_amdgcn_readlane((int)p_base, (int)t);
                const uint32_t qt = (uint32_t)__builtin_amdgcn_readlane((int)p_quota, (int)t);
                if (lane >= bt0 && lane < bt0 + qt) {
                    p_st = t;
                    p_so = lane - bt0;
                }
            }
        };
        auto plan_tile = [&](uint32_t buf) {
            // MaxScore split (search.rs:153-169 is this test, one document at a time): the longest prefix of the
            // terms in ascending upper-bound order whose bounds sum below the threshold is NON-ESSENTIAL -- a
            // document made only of those terms cannot enter the top-k.  Their blocks are not planned at all.
            sv_cur[0] = sv_cur[1];
            sv_tlo[0] = sv_tlo[1];
            sv_cur[1] = p_cur;
            sv_tlo[1] = p_tlo;
            if (++p_good >= 8u) {  // eight tiles without an overflow: try larger tiles again
                p_good = 0;
                p_cap = min(2 * p_cap, (uint32_t)R_NBLK);
            }
            const uint32_t mqp = uni(S.mq);
            const double thd = __longlong_as_double((long long)theta_lds());
            uint32_t p_th = 0;
            for (uint32_t pp = 1; pp <= mqp; ++pp)
                if (S.t_cum[pp] < thd) p_th = pp;
            if (p_th == mqp) {  // no document at all can reach the threshold any more
                if (lane == 0) S.hdr[buf] = make_uint4(p_tlo, p_tlo, 0, 0);
                return;
            }
            // (p_relax: the essential lists intersect too densely even for the smallest tiles -- every term the
            // threshold allows becomes non-essential, whatever the lookups cost)
            const uint32_t p_new = !bt.ne_on ? 0u : p_relax ? p_th : (uint32_t)S.t_best[p_th];
            if (p_new > p_ne) {
                p_ne = p_new;
                assign_quotas();
            }
            const double nesum = S.t_cum[p_ne];
            const bool alive = lane < mqp && p_rank >= p_ne && p_cur < p_end;
            uint32_t bnd = NONE32;
            if (alive && p_cur + p_quota < p_end) bnd = ix.blk_min_doc[p_cur + p_quota];
            if (!__ballot(alive) || p_tlo >= hi) {
                if (lane == 0) S.hdr[buf] = make_uint4(p_tlo, p_tlo, 0, 0);
                return;
            }
            const uint32_t st = p_st < 64u ? p_st : 0u;
            const uint32_t cur_s = (uint32_t)__shfl((int)p_cur, (int)st), end_s = (uint32_t)__shfl((int)p_end, (int)st);
            const uint32_t quo_s = (uint32_t)__shfl((int)p_quota, (int)st);
            const uint32_t j = cur_s + p_so;
            const bool valid = p_st != NONE32 && p_so < quo_s && j < end_s;
            uint4 meta = make_uint4(NONE32, 0, 0, 0);
            double ub = 0.0;
            if (valid) {  // (issued before the boundary above is waited for: one round trip to memory per plan, not two)
                meta = ix.blk_meta[j];
                ub = ix.blk_ub[j];
            }
            uint32_t thi = min(hi, wave_min_u32(bnd));
            // the 64 candidates (quotas) may hold more than a tile takes: the largest thi with <= R_NBLK blocks
            if ((uint32_t)__popcll(__ballot(valid && meta.x < thi)) > p_cap) {
                uint32_t lo_v = p_tlo + 1, hi_v = thi;  // count(lo_v) <= terms <= p_cap < count(hi_v)
                while (hi_v - lo_v > 1) {
                    const uint32_t mid = lo_v + ((hi_v - lo_v) >> 1);
                    if ((uint32_t)__popcll(__ballot(valid && meta.x < mid)) <= p_cap) lo_v = mid;
                    else hi_v = mid;
                }
                thi = lo_v;
            }
            const bool in_tile = valid && meta.x < thi;
            const unsigned long long mask = __ballot(in_tile);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (in_tile) {
                S.pm[buf][pos] = meta;
                S.pa[buf][pos] = make_uint2(j, p_st);
                S.pub[buf][pos] = ub;
                S.pw[buf][pos % (RNW - 1)][pos / (RNW - 1)] = make_uint2(j, meta.x);
            }
            {   // slots beyond the plan: its last block (block 0 for a plan of no blocks); lane = slot
                const uint32_t npl = (uint32_t)__popcll(mask);
                const int lastl = mask ? 63 - (int)__builtin_clzll(mask) : 0;
                const uint32_t jl = mask ? (uint32_t)__builtin_amdgcn_readlane((int)j, lastl) : 0u;
                const uint32_t xl = mask ? (uint32_t)__builtin_amdgcn_readlane((int)meta.x, lastl) : 0u;
                if (lane >= npl && lane < (uint32_t)R_NBLK) S.pw[buf][lane % (RNW - 1)][lane / (RNW - 1)] = make_uint2(jl, xl);
            }
            // cold blocks (search.rs:203): upper bound at or above the threshold -- the threshold only rises, so
            // deciding here, one tile early, errs on the safe side.  Bit i of word w: entry (w - 1) + (RNW - 1) i
            if (lane < (uint32_t)RNW) S.coldw[buf][lane] = 0;
            if (in_tile && thd <= ub + nesum)
                atomicOr(&S.coldw[buf][1u + pos % (RNW - 1)], 1u << (pos / (RNW - 1)));
            const unsigned long long cmask = __ballot(in_tile && meta.y < thi);
            if (lane <= (uint32_t)RT) {  // lane t: entries before term t's slots = first entry of term t
                const unsigned long long below = p_base >= 64u ? ~0ull : ((1ull << p_base) - 1ull);
                S.ptb[buf][lane] = (uint8_t)__popcll(mask & below);
                const unsigned long long qm = p_quota >= 64u ? ~0ull : ((1ull << p_quota) - 1ull);
                if (p_base < 64u) p_cur += (uint32_t)__popcll((cmask >> p_base) & qm);
            }
            // bit 8 of the last word: every block of the tile has its post_rel16 word (the workers then skip the per-block tests:
            // S1 is bound by its scalar instructions -- one issue slot per SIMD every fourth cycle -- before its vector ones)
            // (an index made without the plane -- the `rel16_plane` switch of index creation -- has none: every block is decoded)
            const bool all_rel16 = ix.post_rel16 != nullptr && !__ballot(in_tile && !rel16_block(meta.x, meta.y, meta.w));
            if (lane == 0) S.hdr[buf] = make_uint4(p_tlo, thi, (uint32_t)__popcll(mask), p_ne | (all_rel16 ? 0x100u : 0u));
            p_tlo = thi;
        };
        // ---- item setup (wave 0): terms, cursors, quotas, slot map; the first two plans
        if (wave == 0) {
            poll_request();
            const KernArgsP ca = cold_args();
            uint32_t m = 0, term = NONE32;
            {
                const uint32_t qb = uni(ca->bt.q_off[q]), qe = uni(ca->bt.q_off[q + 1]);
                if (qe - qb <= 64) {  // one load per lane, compaction of the indexed terms through LDS
                    const uint32_t tt = lane < qe - qb ? ca->bt.term_ids[qb + lane] : NONE32;
                    const bool ok = tt < ix.n_terms;  // search.rs:59-61
                    const unsigned long long okm = __ballot(ok);
                    if (ok) S.scratch[__builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = tt;
                    __builtin_amdgcn_wave_barrier();
                    m = (uint32_t)__popcll(okm);
                    uint32_t sl = lane;
                    asm volatile("" : "+v"(sl));  // (the address below is otherwise computed at kernel entry and kept in scratch)
                    if (lane < m) term = S.scratch[sl];
                    __builtin_amdgcn_wave_barrier();
                } else {
                    for (uint32_t p = qb; p < qe; ++p) {
                        const uint32_t tt = ca->bt.term_ids[p];
                        if (tt >= ix.n_terms) continue;
                        if (m == lane) term = tt;
                        ++m;
                    }
                }
            }
            m = uni(m);
            const bool act = lane < m;
            double s0 = 0.0, tub = 0.0, kth = 0.0;
            uint32_t df = 0, b0 = 0, b1 = 0;
            p_cur = p_end = 0;
            const double *kub = ca->ix.term_kth_ub;
            if (act && kub) {  // (the smallest 2^i >= k: at least k documents of the term score that much)
                uint32_t kidx = 0;
                while ((1u << kidx) < k) ++kidx;
                kth = kub[(size_t)term * 9 + kidx];
            }
            if (act) {
                b0 = ca->ix.term_first_block[term];
                b1 = ca->ix.term_first_block[term + 1];
                s0 = ca->ix.term_s0[term];
                df = ca->ix.term_df[term];
                const double wtf = (double)ca->ix.term_wand_tf[term];
                tub = ((wtf * s0) / (wtf + S.s1[ca->ix.term_wand_fn[term]])) * (1.0 + 1e-12);
                // first block whose max_doc >= lo: guess by interpolation, gallop, then bisect
                uint32_t lo_b = b0, hi_b = b1;
                if (lo != 0 && b1 > b0) {
                    uint32_t g = b0 + (uint32_t)((unsigned long long)(b1 - b0) * lo / ix.n_docs);
                    if (g >= b1) g = b1 - 1;
                    if (ix.blk_max_doc[g] < lo) {
                        lo_b = g + 1;
                        for (uint32_t step = 1; lo_b < hi_b; step *= 4) {
                            const uint32_t p = min(lo_b + step - 1, hi_b - 1);
                            if (ix.blk_max_doc[p] < lo) lo_b = p + 1;
                            else {
                                hi_b = p;
                                break;
                            }
                        }
                    } else {
                        hi_b = g;
                        for (uint32_t step = 1; lo_b < hi_b; step *= 4) {
                            const uint32_t p = hi_b - lo_b >= step ? hi_b - step : lo_b;
                            if (ix.blk_max_doc[p] >= lo) hi_b = p;
                            else {
                                lo_b = p + 1;
                                break;
                            }
                        }
                    }
                    while (lo_b < hi_b) {
                        const uint32_t mid = (lo_b + hi_b) >> 1;
                        if (ix.blk_max_doc[mid] < lo) lo_b = mid + 1; else hi_b = mid;
                    }
                }
                p_cur = lo_b;
                p_end = b1;
                uint32_t tl = lane;
                asm volatile("" : "+v"(tl));  // (as sl above: none of these addresses hoisted to kernel entry)
                S.t_s0[tl] = s0;
                S.t_ub[tl] = tub;
                S.t_b0[tl] = b0;
                S.t_b1[tl] = b1;
            }
            // terms in ascending order of their token upper bound; prefix sums; admissible prefixes
            // theta0: the largest, over the terms, of the term's k-th largest block maximum -- a lower bound of the final k-th
            // score before the first posting is read (scores are >= 0: fmax over the lanes)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) kth = fmax(kth, __shfl_xor(kth, o));
            const unsigned long long theta0 = (unsigned long long)__double_as_longlong(kth);
            p_df = df;
            p_rank = 0;
            double sums0 = 0.0;
            unsigned long long sumdf = 0;
            for (uint32_t t = 0; t < m; ++t) {
                const double ubt = readlane_f64(tub, t);
                if (act && (ubt < tub || (ubt == tub && t < lane))) ++p_rank;
                sums0 += readlane_f64(s0, t);
                sumdf += (uint32_t)__builtin_amdgcn_readlane((int)df, (int)t);
            }
            if (act) S.t_rank[lane] = (uint8_t)p_rank;
            {
                // a prefix of p terms is admissible when its shortest list is still R_NE_RATIO times longer
                // than all the essential lists together (else the lookups cost more than the scan they save)
                double cum = 0.0;
                unsigned long long head = 0;
                uint32_t best = 0;
                if (lane == 0) {
                    S.t_cum[0] = 0.0;
                    S.t_best[0] = 0;
                }
                for (uint32_t pp = 0; pp < m; ++pp) {
                    const uint32_t owner = (uint32_t)__ffsll((long long)__ballot(act && p_rank == pp)) - 1u;
                    cum += readlane_f64(tub, owner);
                    const unsigned long long dfo = (uint32_t)__builtin_amdgcn_readlane((int)df, (int)owner);
                    head += dfo;
                    if (pp + 1 < m && (dfo >= (unsigned long long)ca->bt.ne_ratio * (sumdf - head) || dfo * 16ull >= ix.n_docs)) best = pp + 1;
                    if (lane == 0) {
                        S.t_cum[pp + 1] = cum;
                        S.t_best[pp + 1] = (uint8_t)best;
                        S.t_ord[pp] = (uint8_t)owner;
                    }
                }
            }
            p_ne = 0;
            p_relax = 0;
            p_cap = R_NBLK;
            p_good = 0;
            if (lane == 0) S.mq = m;
            __builtin_amdgcn_wave_barrier();
            assign_quotas();
            const double hscale = (double)CUR_HB / sums0;  // score -> histogram bucket: linear in [0, sum of s0)
            if (lane == 0) {
                if (fused_g) {
                    // the records plan_kernel would have made: scan_many_kernel (items this kernel gives up) and merge_kernel
                    // (queries with such an item) of the general route read them
                    Item rec;
                    rec.q = q;
                    rec.doc_lo = lo;
                    rec.doc_hi = hi;
                    rec.m = m;
                    ca->bt.items[item] = rec;
                    if (item == 0) *ca->bt.n_items = n_items;
                    if (item % fused_g == 0) ca->bt.q_item_base[q] = item;
                    if (item + 1 == n_items) ca->bt.q_item_base[q + 1] = n_items;
                }
                S.q = q;
                S.lo = lo;
                S.hi = hi;
                S.mq = m;
                S.fail = 0;
                S.hscale = hscale;
                S.theta = theta0;
                if (theta0) atomicMax(&bt.theta[q], theta0);
                S.nmulti[0] = 0;
                S.nmulti[1] = 0;
            }
            __builtin_amdgcn_wave_barrier();
            poll_consume();
            p_tlo = lo;
            plan_tile(0);
            pl_store();
        }
        __syncthreads();
        const uint32_t mq = uni(S.mq);

        RegTopK<RK> rtop;
        rtop.init();
        unsigned long long published = 0;
        auto theta_now = [&]() -> unsigned long long {
            unsigned long long th = S.theta;
            th = ((unsigned long long)uni((uint32_t)(th >> 32)) << 32) | uni((uint32_t)th);
            return th;
        };
        // offer whole documents to this wave's list; the ones that can enter it are counted in the histogram
        auto offer = [&](bool has, double sc, uint32_t d) {
            const unsigned long long th = theta_now();
            has = has && (unsigned long long)__double_as_longlong(sc) >= th &&
                  (rtop.cnt < k || better(sc, d, rtop.kth_s, rtop.kth_d));
            if (!__ballot(has)) return;
            if (has) {
                const double hb = sc * S.hscale;
                const uint32_t b = hb >= (double)(CUR_HB - 1) ? (uint32_t)(CUR_HB - 1) : (uint32_t)hb;
                atomicAdd(&hrow[b], 1u);
            }
            rtop.offer(has, sc, d, k, lane);
            if (rtop.cnt >= k) {
                const unsigned long long kb = (unsigned long long)__double_as_longlong(rtop.kth_s);
                if (kb > published) {
                    if (lane == 0) {
                        atomicMax(&S.theta, kb);
                        atomicMax(&bt.theta[q], kb);
                    }
                    published = kb;
                }
            }
        };

        // =====================================================================
        // Tile loop.  Wave 0 plans (one tile ahead) and polls; waves 1..7 decode RB blocks each.
        // =====================================================================
        bool failed = false;
        PROF_T(t_loop);
        PROF_ADD(8, t_item, t_loop);
        uint32_t par = 0;
        // one row per document with a second arrival (de-duplicated through the hash set)
        auto insert_row = [&](uint32_t d) {
            uint32_t slot = (d * 0x9E3779B1u) >> (32 - R_HS_LOG2);
            for (uint32_t probes = 0;; ++probes) {
                if (S.nmulti[par] >= (uint32_t)(R_ROWS * 8 / RT) || probes >= (uint32_t)R_HS) {
                    S.fail = 1;
                    break;
                }
                const uint32_t prev = atomicCAS(&S.hkeys[slot], EMPTY, d);
                if (prev == EMPTY) {
                    const uint32_t r = atomicAdd(&S.nmulti[par], 1u);
                    if (r < (uint32_t)(R_ROWS * 8 / RT)) {
                        S.mdoc[par][r] = d;
                        S.mslot[par][r] = (uint16_t)slot;
                    } else {
                        S.fail = 1;
                    }
                    break;
                }
                if (prev == d) break;
                slot = (slot + 1) & (R_HS - 1);
            }
        };

        // the lane's post_rel16 words of its wave's blocks of a planned tile, loaded one tile ahead (right after the barrier
        // that publishes the plan) so that S1 does not start with a round trip to HBM
        uint32_t reln[RB];
        const uint32_t *relp = ix.post_rel16 ? ix.post_rel16 : reinterpret_cast<const uint32_t *>(ix.blob);
        const uint32_t relmul = ix.post_rel16 ? 64u : 0u;
        auto fetch_rel = [&](uint32_t b) {
            // branch-free (a branch per load made the compiler wait for every load at the join): entries beyond the plan
            // read the plan's last block, a plan of no blocks reads block 0; S1 masks those entries
            const uint4 *row = reinterpret_cast<const uint4 *>(&S.pw[b][wave ? wave - 1u : 0u][0]);
            const uint32_t bmax = ix.n_blocks - 1u;
            uint32_t blk[RB];
#pragma unroll
            for (int i = 0; i < RB / 2; ++i) {
                const uint4 two = row[i];  // {block, first document} of two entries
                blk[2 * i] = two.x;
                blk[2 * i + 1] = two.z;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i)  // (a plan of no blocks leaves its rows as they were: clamped; no plane: the blob's first words, unused)
                reln[i] = relp[(unsigned long long)relmul * min(uni(blk[i]), bmax) + lane];
        };
        fetch_rel(0);

        for (uint32_t tile = 0;; ++tile) {
            const uint32_t buf = tile % R_PLAN_RING;
            par = tile & 1u;
            const uint4 hdr = uni4(S.hdr[buf]);
            const uint32_t tlo = hdr.x, thi = hdr.y, np = hdr.z;
            const uint32_t pne = hdr.w & 0xffu;       // non-essential terms: positions 0..pne-1 of t_ord
            if (np == 0) break;
            const uint32_t span = thi - tlo;
            const bool exact = span <= R_BM_EXACT;
            PROF_T(t_a);

            uint32_t nv = 0;  // this wave's entries: (wave - 1) + 7 i < np  <=>  i < nv
            if (wave == 0) {
                // ---- planner: threshold poll, plan of the next tile (read by the others after barrier A).  (Issuing the poll's
                // loads at the end of the previous turn, a tile ahead, was measured twice -- round 3 and, with the theta0
                // bootstrap in place, round 4: 0.3813 against 0.3796 ms -- and is not worth the six registers it holds.)
                poll_request();
                pl_load();
                poll_consume();
                plan_tile((tile + 1) % R_PLAN_RING);
                pl_store();
            } else {
                // ---- S1: decode, stage, mark
                asm volatile("; MARK_S1_BEGIN");
                nv = (np + (RNW - 1) - wave) / (RNW - 1);
                if (nv > (uint32_t)RB) nv = RB;
                uint4 c[RB];
                const bool allfast = (hdr.w & 0x100u) != 0;  // (the planner's test of every block of the tile)
                {
                    const uint4 *row = reinterpret_cast<const uint4 *>(&S.pw[buf][wave - 1u][0]);
#pragma unroll
                    for (int i = 0; i < RB / 2; ++i) {
                        const uint4 two = row[i];
                        c[2 * i].x = uni(two.y);  // the first document is all the plane blocks need here
                        c[2 * i + 1].x = uni(two.w);
                    }
                }
                asm volatile("; MARK_S1_DECODE");
                // ids of the rel16 blocks: min_doc + the two halves of the lane's word (fetched one tile ahead); entries
                // beyond nv are masked below
                uint32_t d0[RB], d1[RB];
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    d0[i] = c[i].x + (reln[i] & 0xffffu);
                    d1[i] = c[i].x + (reln[i] >> 16);
                }
                // An index WITHOUT the post_rel16 plane (the reference's blocks decoded in the kernel, compression.rs:65-92,
                // bitpacking_u32_ordered.rs:222-237): the bit-packed blocks of the wave -- every block but a term's tail -- in groups of
                // four: the groups' words are requested together (two 8-byte loads per lane and block: the lane's two fields and
                // what they may straddle into), then extracted and prefix-summed with DPP shifts.  (Round 5 decoded block after
                // block: a round trip to memory and six through the LDS crossbar each -- 0.75 ms on C3.)
                uint32_t bp_done = 0;  // entries decoded here
                if (!ix.post_rel16) {
                    constexpr int DG = VBM25_DECODE_GROUP < RB ? VBM25_DECODE_GROUP : RB;
#pragma unroll
                    for (int g4 = 0; g4 < RB; g4 += DG) {
                        uint32_t w0[DG], w1[DG], w2[DG], w3[DG], wid[DG], mind[DG];
#pragma unroll
                        for (int u = 0; u < DG; ++u) {
                            const int i = g4 + u;
                            const uint4 cc = uni4(S.pm[buf][(wave - 1u) + (RNW - 1) * ((uint32_t)i < nv ? (uint32_t)i : 0u)]);
                            const uint32_t md = (cc.w >> 8) & 0xff, n = cc.w & 0xff;
                            const bool bp = (uint32_t)i < nv && (md >> 7) == 0 && (md & 127u) < 32u && (md & 127u) != 0 && n == 128u;
                            wid[u] = bp ? md & 127u : 0u;  // (0: not decoded here)
                            mind[u] = cc.x;
                            bp_done |= bp ? 1u << i : 0u;
                            // (an entry that is not decoded here reads its own body's first words; an entry BEYOND the wave's share of the
                            // plan -- a wave without any block of a small tile has nv = 0: its "entry 0" is not an entry of this plan but
                            // whatever the LDS holds there -- reads the blob's: found by the 2^30-document index of tests/test_gpu_codec.py,
                            // where the stale offset pointed outside every allocation)
                            pair_fetch(ix.blob + ((uint32_t)i < nv ? 8ull * cc.z : 0ull), bp ? wid[u] : 1u, lane, w0[u], w1[u], w2[u], w3[u]);
                        }
#pragma unroll
                        for (int u = 0; u < DG; ++u) {
                            const int i = g4 + u;
                            uint32_t v0, v1;
                            pair_extract(wid[u] ? wid[u] : 1u, lane, w0[u], w1[u], w2[u], w3[u], v0, v1);
                            const uint32_t own = v0 + v1;
                            const uint32_t incl = wave_incl_scan_u32(own);
                            const uint32_t a0 = mind[u] + (incl - own) + v0;
                            if (wid[u]) {
                                d0[i] = a0;
                                d1[i] = a0 + v1;
                            }
                        }
                    }
                }
                if (!allfast) {  // wide, raw (width 32) or byte-packed tail blocks: generic, synchronous decode from the blob
#pragma nounroll
                    for (uint32_t i = 0; i < nv; ++i) {
                        if ((bp_done >> i) & 1u) continue;
                        const uint4 cc = uni4(S.pm[buf][(wave - 1u) + (RNW - 1) * i]);
                        const uint32_t md = (cc.w >> 8) & 0xff;
                        if (!ix.post_rel16 || !rel16_block(cc.x, cc.y, cc.w)) {
                            const uint32_t n = cc.w & 0xff;
                            uint32_t a0, a1;
                            decode_doc_ids(ix.blob + 8ull * cc.z, md, n, cc.x, lane, a0, a1);
                            a0 = 2 * lane < n ? a0 : NONE32;
                            a1 = 2 * lane + 1 < n ? a1 : NONE32;
#pragma unroll
                            for (int j = 0; j < RB; ++j) {
                                d0[j] = i == (uint32_t)j ? a0 : d0[j];
                                d1[j] = i == (uint32_t)j ? a1 : d1[j];
                            }
                        }
                    }
                }
                asm volatile("; MARK_S1_MARKS");
                if (lane < 4 * RB) S.done[((wave - 1u) + (RNW - 1) * (lane >> 2)) * 4 + (lane & 3)] = 0;  // last tile's done bits
                // stage + mark.  Out-of-range postings (other tiles' documents, padding, entries beyond nv) mark
                // nothing.
                uint32_t o0[RB], o1[RB], m0[RB], m1[RB];
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const uint32_t e = (wave - 1u) + (RNW - 1) * i;
                    *reinterpret_cast<uint2 *>(&S.stage[e * R_STAGE_STRIDE + 2 * lane]) = make_uint2(d0[i], d1[i]);
                    const uint32_t sp = (uint32_t)i < nv ? span : 0u;
                    const uint32_t x0 = d0[i] - tlo, x1 = d1[i] - tlo;
                    m0[i] = 1u << (x0 & 31);
                    m1[i] = 1u << (x1 & 31);
                    if (!exact) {  // second bit hashed from the whole offset (aliases 2^17 apart get different bits)
                        m0[i] |= 1u << ((x0 * 0x9E3779B1u) >> 27);
                        m1[i] |= 1u << ((x1 * 0x9E3779B1u) >> 27);
                    }
                    if (x0 >= sp) m0[i] = 0;  // out of range: the atomic changes nothing
                    if (x1 >= sp) m1[i] = 0;
                    o0[i] = atomicOr(&S.bm[(x0 >> 5) & (R_BM_WORDS - 1)], m0[i]);
                    o1[i] = atomicOr(&S.bm[(x1 >> 5) & (R_BM_WORDS - 1)], m1[i]);
                }
                asm volatile("; MARK_S1_DUPS");
                uint32_t dupmask = 0;
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    dupmask |= ((m0[i] != 0 && (o0[i] & m0[i]) == m0[i]) ? 1u : 0u) << (2 * i) |
                               ((m1[i] != 0 && (o1[i] & m1[i]) == m1[i]) ? 1u : 0u) << (2 * i + 1);
                if (__ballot(dupmask != 0)) {  // second arrivals -> list -> one insert pass
                    uint32_t mask = dupmask;
                    while (mask) {
                        const uint32_t b = (uint32_t)__ffs((int)mask) - 1u;
                        mask &= mask - 1u;
                        // the id comes back from the wave's own stage row (one LDS read instead of a 16-way select)
                        const uint32_t d = S.stage[((wave - 1u) + (RNW - 1) * (b >> 1)) * R_STAGE_STRIDE + 2 * lane + (b & 1u)];
                        const uint32_t pos = atomicAdd(&S.lcnt[wave], 1u);
                        if (pos < (uint32_t)R_LIST) S.list[wave][pos] = d;
                        else S.fail = 2;
                    }
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t n = min(uni(S.lcnt[wave]), (uint32_t)R_LIST);
                    for (uint32_t j = lane; j < n; j += 64) insert_row(S.list[wave][j]);
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) S.lcnt[wave] = 0;
                }
#ifdef VBM25_PROFILE
                prof[0] += 1;
#endif
            }
            asm volatile("; MARK_S1_END");
            PROF_T(t_b);
            PROF_ADD(1, t_a, t_b);
            lds_barrier();  // ---- A: every mark, staged id and row of the tile is in LDS; the next plan too
            PROF_T(t_c);
            PROF_ADD(2, t_b, t_c);

            if (uni(S.fail)) {
                // The tile overflowed its rows (or a wave its list of second arrivals): lists that intersect that
                // densely take smaller tiles.  Undo the tile and plan it again with half the blocks as the NEXT
                // tile; an item that overflows even at one block per term goes to scan_many_kernel.
                lds_barrier();  // everybody has seen the flag
                for (uint32_t i = tid; i < R_HS; i += RWG) S.hkeys[i] = EMPTY;
                for (uint32_t i = tid; i < R_BM_WORDS; i += RWG) S.bm[i] = 0;
                if (tid < RNW) S.lcnt[tid] = 0;
                if (wave == 0) {
                    pl_load();
                    bool give_up = false;
                    const uint32_t n_ess = uni(S.mq) - p_ne;  // a tile holds at least one block per essential term
                    if (p_cap <= n_ess) {  // smallest tiles already: relax the admissibility rule once
                        give_up = give_up || p_relax != 0;
                        p_relax = 1;
                    }
                    if (lane == 0) {
                        S.fail = give_up ? (0x10u | S.fail | p_ne << 8 | min(S.nmulti[par], 255u) << 16 | (exact ? 1u << 24 : 0u) | min(np, 127u) << 25) : 0u;
                        S.nmulti[0] = 0;
                        S.nmulti[1] = 0;
                    }
                    if (!give_up) {
                        p_cap = max(p_cap >> 1, n_ess);
                        p_good = 0;
                        p_cur = sv_cur[0];
                        p_tlo = sv_tlo[0];
                        sv_cur[1] = p_cur;
                        sv_tlo[1] = p_tlo;
                        plan_tile((tile + 1) % R_PLAN_RING);  // replaces the plan made from the failed tile's end
                    }
                    pl_store();
                }
                lds_barrier();
                if (uni(S.fail)) {
                    failed = true;
                    break;
                }
                fetch_rel((tile + 1) % R_PLAN_RING);
                continue;  // the next iteration runs the re-planned tile (a backward goto into this loop cost 20 %)
            }
            asm volatile("; MARK_PREFETCH");
            fetch_rel((tile + 1) % R_PLAN_RING);
            asm volatile("; MARK_S2_BEGIN");
            const uint32_t nm = min(uni(S.nmulti[par]), (uint32_t)(R_ROWS * 8 / RT));

            // ---- S2: wipe the filter; rows x terms: find the postings, score them
#pragma unroll
            for (int i = 0; i < (R_BM_WORDS / 4 + RWG - 1) / RWG; ++i)
                if (R_BM_WORDS / 4 % RWG == 0 || tid + i * RWG < (uint32_t)(R_BM_WORDS / 4))
                    reinterpret_cast<uint4 *>(S.bm)[tid + i * RWG] = make_uint4(0, 0, 0, 0);
            if (tid == 0) S.nmulti[par ^ 1u] = 0;
            if (nm) {
                // tasks = rows x the query's terms, packed: waves beyond nm * mq tasks skip the phase
                const uint32_t inv_mq = (65536u + mq - 1u) / mq;  // p / mq == (p * inv_mq) >> 16 for p < 4096
                for (uint32_t p = tid; p < nm * mq; p += RWG) {
                    const uint32_t r = (p * inv_mq) >> 16, t = p - r * mq;
                    const uint32_t d = S.mdoc[par][r];
                    if (t == 0) S.hkeys[S.mslot[par][r]] = EMPTY;
                    uint32_t eb = S.ptb[buf][t], len = S.ptb[buf][t + 1] - eb;
                    if (len == 0) continue;
                    while (len > 1) {  // last entry of the term with min_doc <= d
                        const uint32_t half = len >> 1;
                        if (S.pm[buf][eb + half].x <= d) {
                            eb += half;
                            len -= half;
                        } else {
                            len = half;
                        }
                    }
                    const uint4 sj = S.pm[buf][eb];
                    if (d < sj.x || d > sj.y) continue;
                    const uint32_t *sb = &S.stage[eb * R_STAGE_STRIDE];
                    uint32_t idx = 0;
#pragma unroll
                    for (int s = 64; s > 0; s >>= 1)
                        if (sb[idx + s - 1] < d) idx += s;
                    if (sb[idx] != d) continue;
                    atomicOr(&S.done[eb * 4 + (idx >> 5)], 1u << (idx & 31));
                    double tf;
                    uint32_t fn;
#ifdef VBM25_S2_NOLOAD  // timing experiment only (wrong scores): what S2 costs without its round trip to HBM
                    tf = (double)((idx & 3u) + 1u);
                    fn = idx & 255u;
#else
                    if (tfn_block(sj.w)) {  // term frequency and fieldnorm of the posting from ONE word of the derived plane
                        const uint32_t ww = ix.post_tfn[64ull * S.pa[buf][eb].x + (idx >> 1)] >> ((idx & 1u) * 8u);
                        tf = (double)(ww & 0xffu);
                        fn = (ww >> 16) & 0xffu;
                    } else {  // tails, tf fields wider than 7 bits: the field out of the blob, the fieldnorm from its own plane
                        const uint32_t nj = sj.w & 0xff, mdj = (sj.w >> 8) & 0xff, mtj = (sj.w >> 16) & 0xff;
                        const uint8_t *tbody = ix.blob + 8ull * sj.z + ((payload_bytes(mdj, nj) + 7u) & ~7u);
                        const FieldAddr fa = field_addr(mtj, nj, idx);
                        const uint32_t flo = *reinterpret_cast<const uint32_t *>(tbody + fa.off0);
                        const uint32_t fhi = *reinterpret_cast<const uint32_t *>(tbody + fa.off1);
                        fn = ix.post_fn[128ull * S.pa[buf][eb].x + idx];
                        tf = (double)field_val(flo, fhi, fa);
                    }
#endif
                    S.contrib[(r << LRT) + t] = (tf * S.t_s0[t]) / (tf + S.s1[fn]);  // Cache::evaluate, bm25.rs:355-358
                }
            }
            asm volatile("; MARK_S2_END");
            PROF_T(t_d);
            PROF_ADD(5, t_c, t_d);
            lds_barrier();  // ---- B: contributions and done bits complete; filter clean
            PROF_T(t_e);
            PROF_ADD(6, t_d, t_e);
#ifdef VBM25_PROFILE
            prof[10] += nm;
#endif

            // ---- completion of a candidate by lookups in the non-essential lists (all 64 lanes call; `cand` marks
            // the lanes that hold one): partial = its score over the essential terms, which come from the row
            // (row != NONE32) or are the single posting (tself, pself).
            const double nesum = pne ? S.t_cum[pne] : 0.0;
            uint32_t emask = 0xffffffffu;  // bit t: term t is essential
            if (pne) {
                emask = 0;
                for (uint32_t t = 0; t < mq; ++t) emask |= ((uint32_t)S.t_rank[t] >= pne ? 1u : 0u) << t;
            }
            // em: the essential terms (their contributions are known), nes: sum of the other terms' token bounds
            auto complete = [&](bool cand, uint32_t d, uint32_t row, uint32_t tself, double pself, double partial,
                                uint32_t em, double nes) {
                const double thd = __longlong_as_double((long long)theta_now());
                cand = cand && partial + nes >= thd;
                if (!__ballot(cand)) return;
                // pass 1: block upper bounds (search.rs:177-203) instead of the token bounds
                double bound = partial;
                for (uint32_t t = 0; t < mq; ++t) {
                    if ((em >> t) & 1u) continue;
                    const uint32_t b1 = S.t_b1[t];
                    if (cand) {
                        const uint32_t b = r_first_block_ge(ix, S.t_b0[t], b1, d);
                        if (b < b1 && ix.blk_min_doc[b] <= d) bound += ix.blk_ub[b];
                    }
                }
                cand = cand && bound * (1.0 + 1e-12) >= thd;
                if (!__ballot(cand)) return;
                // pass 2: the exact score, terms in ascending key order (evaluate.rs:43-72)
                uint32_t *scr = S.list[wave];  // 128 ids of the block being looked into
                double acc = 0.0;
                for (uint32_t t = 0; t < mq; ++t) {
                    double c = 0.0;
                    if ((em >> t) & 1u) {
                        if (cand) c = row != NONE32 ? S.contrib[(row << LRT) + t] : (t == tself ? pself : 0.0);
                    } else {
                        const uint32_t b1 = S.t_b1[t];
                        uint32_t b = NONE32;
                        bool pend = false;
                        if (cand) {
                            b = r_first_block_ge(ix, S.t_b0[t], b1, d);
                            pend = b < b1 && ix.blk_min_doc[b] <= d;
                        }
                        for (;;) {
                            const unsigned long long pmask = __ballot(pend);
                            if (!pmask) break;
                            const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)b, __ffsll((long long)pmask) - 1);
                            const uint4 bm = uni4(ix.blk_meta[blk]);
                            const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
                            uint32_t a0, a1;
                            decode_doc_ids(ix.blob + 8ull * bm.z, md, n, bm.x, lane, a0, a1);
                            __builtin_amdgcn_wave_barrier();
                            *reinterpret_cast<uint2 *>(&scr[2 * lane]) = make_uint2(2 * lane < n ? a0 : NONE32, 2 * lane + 1 < n ? a1 : NONE32);
                            __builtin_amdgcn_wave_barrier();
                            if (pend && b == blk) {
                                uint32_t idx = 0;
#pragma unroll
                                for (int sft = 64; sft > 0; sft >>= 1)
                                    if (scr[idx + sft - 1] < d) idx += sft;
                                if (scr[idx] == d) {
                                    const uint8_t *tbody = ix.blob + 8ull * bm.z + ((payload_bytes(md, n) + 7u) & ~7u);
                                    const FieldAddr fa = field_addr(mt, n, idx);
                                    const uint32_t flo = *reinterpret_cast<const uint32_t *>(tbody + fa.off0);
                                    const uint32_t fhi = *reinterpret_cast<const uint32_t *>(tbody + fa.off1);
                                    const uint32_t fn = ix.post_fn[128ull * blk + idx];
                                    const double tf = (double)field_val(flo, fhi, fa);
                                    c = (tf * S.t_s0[t]) / (tf + S.s1[fn]);
                                }
                                pend = false;
                            }
                        }
                    }
                    acc += c;
                }
                offer(cand, acc, d);
            };

            // ---- S3: rows -> documents.  Wave w takes the rows [w R, (w + 1) R): consecutive lanes read consecutive
            // rows of contrib (a row stride of RNW rows put every lane on the same LDS banks), and the waves
            // beyond the last row skip the phase
            constexpr uint32_t RPW = (uint32_t)(R_ROWS * 8 / RT) / RNW;
            if (nm > wave * RPW) {
                const uint32_t r = wave * RPW + lane;
                const bool has = lane < RPW && r < nm;
                double acc = 0.0;
                uint32_t d = 0;
                if (has) {
                    d = S.mdoc[par][r];
                    for (uint32_t t = 0; t < mq; ++t) acc += S.contrib[(r << LRT) + t];  // ascending key order; absent terms add 0.0
                }
                if (pne == 0) offer(has, acc, d);
                else complete(has, d, has ? r : NONE32, 0, 0.0, acc, emask, nesum);
                if (has)
                    for (uint32_t t = 0; t < mq; ++t) S.contrib[(r << LRT) + t] = 0.0;
            }
            PROF_T(t_f);
            PROF_ADD(7, t_e, t_f);
            // ---- cold pass: blocks whose upper bound reaches the threshold (search.rs:203)
            {
                if (wave != 0) {
                    uint32_t coldmask = uni(S.coldw[buf][wave]);
                    while (coldmask) {
                        const uint32_t i = (uint32_t)__ffs((int)coldmask) - 1u;
                        coldmask &= coldmask - 1u;
                        const uint32_t e = (wave - 1u) + (RNW - 1) * i;
                        if (e >= np) continue;
                        {   // the planner decided one tile early: check against the threshold of now
                            const unsigned long long ubb = (unsigned long long)__double_as_longlong(S.pub[buf][e]);
                            const unsigned long long ubu = ((unsigned long long)uni((uint32_t)(ubb >> 32)) << 32) | uni((uint32_t)ubb);
                            if (__longlong_as_double((long long)theta_now()) > __longlong_as_double((long long)ubu) + nesum) continue;
                        }
                        // ids from this wave's own stage row (nobody else writes it)
                        const uint2 dd = *reinterpret_cast<const uint2 *>(&S.stage[e * R_STAGE_STRIDE + 2 * lane]);
                        const uint32_t dwi = S.done[e * 4 + (lane >> 4)];
                        const bool ok0 = dd.x - tlo < span && !((dwi >> ((2 * lane) & 31)) & 1u);
                        const bool ok1 = dd.y - tlo < span && !((dwi >> ((2 * lane + 1) & 31)) & 1u);
                        if (__ballot(ok0 || ok1)) {
                            const uint4 sj = uni4(S.pm[buf][e]);
                            const uint2 aux = S.pa[buf][e];
                            const uint32_t blkj = uni(aux.x), t = uni(aux.y);
                            double tf0, tf1;
                            uint32_t fnp;  // the two fieldnorm bytes
                            if (tfn_block(sj.w)) {  // (uniform: one block) the lane's two postings in ONE word of the derived plane
                                const uint32_t ww = ix.post_tfn[64ull * blkj + lane];
                                tf0 = (double)(ww & 0xffu);
                                tf1 = (double)((ww >> 8) & 0xffu);
                                fnp = ww >> 16;
                            } else {
                                const uint32_t nj = sj.w & 0xff, mdj = (sj.w >> 8) & 0xff, mtj = (sj.w >> 16) & 0xff;
                                const uint8_t *tbody = ix.blob + 8ull * sj.z + ((payload_bytes(mdj, nj) + 7u) & ~7u);
                                const FieldAddr f0 = field_addr(mtj, nj, 2 * lane), f1 = field_addr(mtj, nj, 2 * lane + 1);
                                const uint32_t l0 = *reinterpret_cast<const uint32_t *>(tbody + f0.off0);
                                const uint32_t h0 = *reinterpret_cast<const uint32_t *>(tbody + f0.off1);
                                const uint32_t l1 = *reinterpret_cast<const uint32_t *>(tbody + f1.off0);
                                const uint32_t h1 = *reinterpret_cast<const uint32_t *>(tbody + f1.off1);
                                fnp = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * blkj)[lane];
                                tf0 = (double)field_val(l0, h0, f0);
                                tf1 = (double)field_val(l1, h1, f1);
                            }
                            const double s0t = S.t_s0[t];
                            const double p0 = (tf0 * s0t) / (tf0 + S.s1[fnp & 0xff]);
                            const double p1 = (tf1 * s0t) / (tf1 + S.s1[fnp >> 8]);
#pragma nounroll
                            for (uint32_t si = 0; si < 2; ++si) {  // the two postings of the lane, one call site each
                                const bool ok = si ? ok1 : ok0;
                                const double pp = si ? p1 : p0;
                                const uint32_t dx = si ? dd.y : dd.x;
                                if (pne == 0) offer(ok, pp, dx);
                                else complete(ok, dx, NONE32, t, pp, pp, emask, nesum);
                            }
    #ifdef VBM25_PROFILE
                            prof[11] += 1;
    #endif
                        }
                    }
                }
            }
            PROF_T(t_g);
            PROF_ADD(13, t_f, t_g);
        }

#ifdef VBM25_PROFILE
        prof[12] += 1;
        prof[9] += __builtin_readcyclecounter() - t_loop;
#endif
        // ---- item result: one list per wave
        const uint32_t n = failed ? 0u : rtop.cnt;
        const KernArgsP ce = cold_args();
        const size_t list = (size_t)item * ce->bt.lpi + wave;
        {
            double *res_score = ce->bt.res_score;
            uint32_t *res_doc = ce->bt.res_doc;
#pragma unroll
            for (int r = 0; r < RK; ++r)
                if (r * 64 + lane < n) {
                    res_score[list * k + r * 64 + lane] = rtop.score[r];
                    res_doc[list * k + r * 64 + lane] = rtop.doc[r];
                }
        }
        if (lane == 0) {
            ce->bt.res_cnt[list] = n;
            if (wave == 0) {
                ce->bt.item_failed[item] = failed ? (S.fail | 0x100u) : 0u;
                if (failed) *ce->bt.fail_any = 1u;
            }
        }
        if constexpr (FUSED) {
            // ---- the last workgroup of the query merges its lists (merge.h's job) and leaves the per-launch state clean.
            // One item per query: the eight lists are this workgroup's own and travel through LDS.  Several items: an
            // atomic counter per query finds the last workgroup; the other workgroups' lists are read with loads that
            // bypass this CU's vector cache, all entries of 16 lists at a time (one round trip, not one per list).
            uint32_t *tmp = S.stage;  // (free between items)
            bool last = true;
            // (ml, zero: the lane number and a zero made opaque -- addresses built from the lane number in this block, and the
            // zeros it stores, are otherwise computed at kernel entry and kept in scratch over the whole launch)
            uint32_t ml = lane, zero = 0;
            asm volatile("" : "+v"(ml), "+v"(zero));
            if (fused_g == 1u) {
                __syncthreads();  // every wave is done with its stage rows (the cold pass of the last tile reads them)
                const uint32_t base = wave * 3u * (uint32_t)KMAX;
#pragma unroll
                for (int r = 0; r < RK; ++r)
                    if (r * 64 + lane < n) {
                        tmp[base + r * 64 + ml] = (uint32_t)__double2loint(rtop.score[r]);
                        tmp[base + KMAX + r * 64 + ml] = (uint32_t)__double2hiint(rtop.score[r]);
                        tmp[base + 2 * KMAX + r * 64 + ml] = rtop.doc[r];
                    }
                if (lane == 0) S.lcnt[wave] = n;
                __syncthreads();
            } else {
                __threadfence();
                __syncthreads();
                if (tid == 0) S.scratch[0] = atomicAdd(&bt.fused_state[1 + q], 1u);
                __syncthreads();
                last = uni(S.scratch[0]) == fused_g - 1u;
            }
            if (last && wave == 0) {
                rtop.init();
                uint32_t any_failed = failed ? 1u : 0u;
                const uint32_t i0 = q * fused_g;
                if (fused_g == 1u) {
                    for (uint32_t w = 0; w < (uint32_t)RNW; ++w) {
                        const uint32_t cnt = uni(S.lcnt[w]), base = w * 3u * (uint32_t)KMAX;
                        for (uint32_t e0 = 0; e0 < cnt; e0 += 64) {
                            const bool has = e0 + lane < cnt;
                            double sc = 0;
                            uint32_t d = 0;
                            if (has) {
                                sc = __hiloint2double((int)tmp[base + KMAX + e0 + lane], (int)tmp[base + e0 + lane]);
                                d = tmp[base + 2 * KMAX + e0 + lane];
                            }
                            rtop.offer(has, sc, d, k, lane);
                        }
                    }
                } else {
                    __threadfence();
                    const uint32_t L0 = i0 * bt.lpi, NL = fused_g * bt.lpi;
                    for (uint32_t lb = 0; lb < NL; lb += 16) {
                        uint32_t cnt = 0;
                        if (lane < 16u && lb + lane < NL) cnt = min(__hip_atomic_load(&bt.res_cnt[L0 + lb + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), k);
                        const uint32_t incl = wave_incl_scan_u32(cnt), excl = incl - cnt;
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);  // <= 16 k <= 4096 words of the stage
                        for (uint32_t j = 0; j < cnt; ++j) tmp[excl + j] = lane << 16 | j;
                        __builtin_amdgcn_wave_barrier();
                        for (uint32_t e0 = 0; e0 < total; e0 += 64) {
                            const bool has = e0 + lane < total;
                            double sc = 0;
                            uint32_t d = 0;
                            if (has) {
                                const uint32_t ds = tmp[e0 + lane];
                                const size_t at = (size_t)(L0 + lb + (ds >> 16)) * k + (ds & 0xffffu);
                                sc = __longlong_as_double((long long)__hip_atomic_load(
                                    reinterpret_cast<unsigned long long *>(&bt.res_score[at]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                d = __hip_atomic_load(&bt.res_doc[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            rtop.offer(has, sc, d, k, lane);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (fused_g != 1u)
                    for (uint32_t i = lane; i < fused_g; i += 64)
                        any_failed |= __hip_atomic_load(&bt.item_failed[i0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t nh = rtop.cnt;
                const bool failed_any = __ballot(any_failed != 0) != 0ull;
#pragma unroll
                for (int r = 0; r < RK; ++r)
                    if (r * 64 + lane < nh) {
                        const uint32_t d = rtop.doc[r];
                        const uint16_t *pl = ix.doc_payload + 3ull * d;
                        unsigned long long *out = reinterpret_cast<unsigned long long *>(bt.hits + (size_t)q * k + r * 64 + ml);
                        out[0] = (unsigned long long)__double_as_longlong(rtop.score[r]);
                        out[1] = (unsigned long long)d | (unsigned long long)pl[0] << 32 | (unsigned long long)pl[1] << 48;
                        out[2] = (unsigned long long)pl[2];
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) hrow[4 * ml + i] = zero;
                if (lane == 0) {
                    bt.n_hits[q] = failed_any ? NONE32 : nh;  // NONE32: an item needs scan_many_kernel -- the host re-runs the batch on the general route
                    bt.theta[q] = 0;
                    if (fused_g != 1u) bt.fused_state[1 + q] = 0;
                }
            }
        }
    }
    if constexpr (FUSED) {  // the last workgroup to leave resets the item counter
        __threadfence();
        __syncthreads();
        if (tid == 0 && atomicAdd(&bt.fused_state[0], 1u) == gridDim.x - 1u) {
            *bt.work_ctr = 0;
            bt.fused_state[0] = 0;
        }
    }
#ifdef VBM25_PROFILE
    if (bt.prof && lane == 0) {
        unsigned long long *o = bt.prof + ((size_t)blockIdx.x * RNW + wave) * 16;
        for (int i = 0; i < 16; ++i) o[i] = prof[i];
        o[15] = __builtin_readcyclecounter() - prof_t0;
    }
#endif
}
