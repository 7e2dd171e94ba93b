// scan_range.h -- scan_range_kernel: the doc-range formulation for sparse queries (<= RT indexed terms, k <= REG_K).
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25.
//
// One 8-wave workgroup per work item (query x doc range), persistent, items from an atomic counter.  The item is cut
// into TILES: a tile is a doc range [tlo, thi) that holds at most R_NBLK posting blocks of the query's terms (every
// block whose min_doc < thi; per-term quotas proportional to df make thi).  Wave 0 is the CONTROL wave: it plans tile
// n + 1 while the others work on tile n, polls the query's shared threshold and sums the rows of tile n - 1; waves
// 1..7 are WORKERS with RB blocks each.  A tile is
//
//   S1  (workers) the raw id words of the wave's blocks are already in LDS -- fetched by LDS-DMA (global_load_lds,
//       no registers, issued one tile ahead into the wave's own slots) -- and are decoded branch-free, eight blocks at
//       a time: v_alignbit field extraction, DPP prefix sum.  The ids are STAGED in LDS relative to the block's first
//       id (16 bits per id; blocks spanning >= 2^16 documents take two slots of 32-bit ids) and every posting is marked
//       in the tile's SEEN filter with ONE returning 32-bit LDS atomic: word (x >> 5) mod 4096, bit x mod 32, plus a
//       second bit of the same word hashed from x >> 17 (x = id - tlo; a blocked Bloom filter that is exact for tiles
//       <= 2^17 documents wide).  A mark that was already there = a SECOND ARRIVAL: the document may sit in two lists.
//       Second arrivals go through a per-wave list into a hash set: one ROW per document.
//   --- barrier A ---
//   S2  (all waves) lanes = (row, term): binary search of the row's document in the term's staged blocks.  A posting
//       found gets its done bit and a HIT record; its tf / fieldnorm bytes are fetched one tile later, all hits of a
//       wave in ONE batch of loads (no memory round trip on the tile's critical path), and Cache::evaluate
//       (bm25.rs:355-358) lands in contrib[row][term].  The workers' LDS-DMA of the next tile is issued here.
//   S3  (wave 0, one tile later) row sums in ascending key order (evaluate.rs:43-72; absent terms add 0.0, exact) ->
//       the item's candidate POOL.
//   --- barrier B ---
//   cold pass (workers, own blocks): a block whose upper bound (search.rs:377-380, evaluated once per index) reaches the
//       threshold has every posting without a done bit scored on its own -> pool; every other block never has its
//       tf / fieldnorm bytes read.
//
// The POOL (LDS, unsorted) replaces a register top-k per wave: a candidate at or above the threshold is appended with
// one LDS atomic; when the pool fills up it is SHRUNK by a histogram selection over the scores (three levels of 256
// buckets) that also raises the threshold; at the item's end the waves cut it into eight sorted lists (RegTopK, the
// only place it is instantiated) for merge_kernel.  The threshold is shared as before: LDS word, the query's 64-bit
// atomicMax word, the 256-bucket histogram of accepted documents.
//
// MaxScore split (search.rs:153-169): terms in ascending order of their token upper bound; the longest admissible
// prefix whose bounds sum below the threshold is NON-ESSENTIAL -- its blocks are neither planned nor fetched, and a
// candidate whose partial score + the non-essential bounds still reaches the threshold is COMPLETED by lookups
// (block by interpolation + gallop + bisection of blk_max_doc, block upper bounds first, search.rs:177-203).
//
// Overflows do not undo a tile: a document that finds no free row goes to the LATE list and is scored by lookups in
// every list (exact, independent of the tile); the planner halves the tile size while rows run short.  An item whose
// late list or pool cannot be brought down goes to scan_many_kernel (item_failed).
//
// Instruction budget (tools/ubench/valu_rates.hip: v_add/sub/and/or/xor/lshr/mov issue in 2.5 cycles per wave, every
// other VALU op incl. v_lshl, v_alignbit, v_cmp, v_cndmask, DPP in 4.3): S1 is written for VALU cycles -- no range
// test per posting (postings of a straddling block that belong to the neighbour tile mark a bit and are filtered on
// the rare paths), uniform values read from LDS into VGPRs instead of SGPRs, no select chains.

constexpr int RNW = 8;               // waves per workgroup: control wave + 7 workers
constexpr int RWG = RNW * 64;
constexpr int RB = 7;                // blocks per worker per tile (the 64 candidate lanes of the planner fill 49 entries almost always)
constexpr int R_NBLK = (RNW - 1) * RB;  // entries per tile (slots = lanes 0..R_NBLK-1 of the control wave)
static_assert(R_NBLK <= 64, "one planner lane per block of a tile");
constexpr int R_BM_WORDS = 4096;     // 2^17 bits
constexpr int R_HS_LOG2 = 9;
constexpr int R_HS = 1 << R_HS_LOG2;  // slots of the second-arrival hash set (rows <= 128: never more than a quarter full + late keys)
constexpr uint32_t R_TARGET_ITEMS = 1024;
constexpr uint32_t R_MIN_CHUNK_POSTINGS = 16384;
constexpr uint32_t R_GRID = 512;      // persistent workgroups: 256 CUs x 2
constexpr int R_PLAN_RING = 3;        // plans: tile n - 1 (hits being scored), n, n + 1 (being planned)
constexpr int R_HITS = 128;           // hit records per wave per tile: rows x terms <= 1024 tasks, two rounds of 64 per wave
constexpr int R_EVENTS = R_HITS;      // per-wave list of second arrivals (the same words; emptied into the hash set whenever it is half full)

constexpr int R_XROWS = 64;           // documents beyond a tile's rows: found (done bits) but scored by lookups (late list)
constexpr int R_LATE = 128;           // documents waiting for the lookup path
constexpr int R_STAGE_SLOTS = R_NBLK;  // staging: slots of 128 x 16 bits; a wide block takes two
constexpr int R_SS = 66;               // words between slots: 64 would put the same column of every row on one LDS bank (S2 probes columns)

template <int KMAX>
struct RangePool {
    static constexpr int N = KMAX <= 64 ? 512 : (KMAX <= 128 ? 640 : 704);
};

template <int KMAX, int RT>
struct RangeLds {
    static constexpr int ROWS = 1024 / RT;      // rows (documents with a second arrival) per tile
    static constexpr int POOL = RangePool<KMAX>::N;
    uint32_t bm[R_BM_WORDS];
    uint4 raw[R_NBLK * 16];            // LDS-DMA target: 256 bytes of id words per entry (full blocks of width <= 15)
    uint32_t stage[R_STAGE_SLOTS * R_SS];  // staged ids: slot = 64 words = 128 x u16 (+ 2 words of skew); wide block: two slots = 128 x u32
    uint32_t hkeys[R_HS];
    uint32_t mdoc[2][ROWS + R_XROWS];  // row -> document, by tile parity
    double contrib[ROWS * RT];
    unsigned long long pool_s[POOL];   // score bits
    uint32_t pool_d[POOL];
    uint32_t hits[RNW][R_HITS];        // row << 16 | entry << 8 | index in block; during S1 (the hit records are read by then)
                                       // the wave's list of second arrivals
    uint32_t done[R_NBLK * 4];
    uint4 pm[R_PLAN_RING][R_NBLK];     // {min_doc, max_doc, off8, n | md << 8 | mt << 16 | wand_fn << 24}
    uint2 pa[R_PLAN_RING][R_NBLK];     // {block index, term | stage slot << 8 | wide << 16 | fast << 17 | valid << 18}
    float pub[R_PLAN_RING][R_NBLK];    // block upper bound, rounded up
    uint32_t coldw[R_PLAN_RING][RNW];  // per worker: its entries whose upper bound reaches the threshold
    uint4 hdr[R_PLAN_RING];            // {tlo, thi, entries, non-essential terms | any cold << 8 | cold blocks << 16}
    uint8_t ptb[R_PLAN_RING][RT + 4];  // first plan entry of each term (entries of a term are contiguous)
    double s1[256];
    double t_s0[RT];
    double t_ub[RT];                   // token upper bound x (1 + 1e-12)
    double t_cum[RT + 1];              // sum of the p smallest token upper bounds
    uint8_t t_rank[RT];                // position of the term in ascending upper-bound order
    uint8_t t_ord[RT];                 // ... and the term at a position
    uint8_t t_best[RT + 1];            // largest admissible non-essential prefix <= p
    uint32_t t_b0[RT], t_b1[RT];       // block range of the term
    uint32_t late[R_LATE];
    double hscale;
    unsigned long long theta;          // bits of a lower bound of the query's k-th best score
    double sel_lo, sel_mul;            // pool selection: focus of the current level
    uint32_t sel_need, sel_b, sel_above, sel_stop;
    uint32_t nrows[3];                 // rows of a tile, by tile mod 3 (mdoc: by parity)
    uint32_t hcnt[RNW];
    uint32_t pool_n, pool_w, nlate, cold_retry, rows_seen, pool_snap, late_snap, theta_zero, plan_seq;
    uint32_t item, q, lo, hi, mq, fail;
    uint32_t scratch[64];
    // planner state (wave 0), kept here between its turns so that the workers do not carry it in registers:
    // per lane {cur, quota, base, slot term, slot offset}, then the uniform words {ne, relax, cap, tlo, prefetch valid}
    uint32_t pl[5][64];
    uint32_t t_df[RT];
    // inputs of the next plan, requested by LDS-DMA at the end of the previous one: candidate lane -> block metadata and upper
    // bound; term lane -> min_doc of the first block beyond its quota
    uint4 pf_meta[64];
    uint32_t pf_ub[2][64];
    uint32_t pf_bnd[64];
    uint32_t plu[5];
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

// LDS-DMA: the active lanes' dwords at gbase + voff land at lds_dst + 4 * lane (lds_dst wave-uniform).  Invisible to
// the compiler's vmcnt bookkeeping: the consumer waits with its own s_waitcnt vmcnt(0).  M0 (the LDS destination) is
// written and restored in the same statement.
__device__ __forceinline__ void r_glds_dword(const uint8_t *gbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}

// ... per-lane source addresses (vaddr form): 16 bytes / 4 bytes per lane at lds_dst + 16 / 4 * lane
__device__ __forceinline__ void r_glds_dwordx4_v(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void r_glds_dword_v(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int KMAX, int RT, bool FUSED = false>
#ifndef R_LB_V
#define R_LB_V 4  // (tools: -DR_LB_V=2 compiles without register pressure)
#endif
__global__ void __launch_bounds__(RWG, R_LB_V) scan_range_kernel(DevIndex ix, DevBatch bt) {
    static_assert(KMAX <= REG_K, "register top-k only");
    static_assert(RT == 8 || RT == 16, "row stride");
    constexpr int RK = KMAX / 64;
    constexpr int LRT = RT == 8 ? 3 : 4;
    using Lds = RangeLds<KMAX, RT>;
    constexpr uint32_t ROWS = Lds::ROWS, POOL = Lds::POOL;
    constexpr uint32_t POOL_COLD = POOL - ROWS;        // the cold pass leaves room for one tile's rows
    constexpr uint32_t POOL_HIGH = POOL_COLD - 128u;   // above this after a tile: shrink (a whole block fits below POOL_COLD afterwards)
    static_assert(sizeof(Lds) <= 81920, "two workgroups per CU");
    static_assert(POOL_HIGH >= (uint32_t)KMAX + 64u, "a shrunk pool must hold k entries and their ties");
    __shared__ Lds S;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t k = bt.k;
    // bt.fused_g != 0 (a handful of queries through vbm25_search_batch): no plan_kernel and no merge_kernel -- every query is
    // cut into fused_g equal document ranges right here, the last workgroup to finish a query merges its lists into the
    // hits and leaves the per-launch state (threshold, histogram, counters) clean for the next launch.
    const uint32_t fused_g = FUSED ? bt.fused_g : 0u;
    const uint32_t n_items = fused_g ? bt.nq * fused_g : *bt.n_items;
    for (uint32_t i = tid; i < 256; i += RWG) S.s1[i] = ix.s1[i];

#ifdef VBM25_PROFILE
    // per wave: 0 tiles, 1 S1, 2 wait A, 3 S2, 4 wait B, 5 cold, 6 hits, 7 control wave: plan, 8 item setup, 9 tile loop,
    // 10 rows, 11 cold blocks, 12 items, 13 hits scored, 14 shrinks, 15 lifetime
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif

    for (;;) {
        __syncthreads();  // previous item fully done with LDS
        if (tid == 0) {
            S.item = atomicAdd(bt.work_ctr, 1u);
            S.nrows[0] = 0;
            S.nrows[1] = 0;
            S.nrows[2] = 0;
            S.pool_n = 0;
            S.nlate = 0;
            S.cold_retry = 0;
            S.rows_seen = 0;
            S.plan_seq = 0;
            S.pool_snap = 0;
            S.late_snap = 0;
            S.theta_zero = 1;
            S.fail = 0;
            S.theta = 0;
        }
        for (uint32_t i = tid; i < R_HS; i += RWG) S.hkeys[i] = EMPTY;
        for (uint32_t i = tid; i < R_BM_WORDS; i += RWG) S.bm[i] = 0;
        for (uint32_t i = tid; i < ROWS * RT; i += RWG) S.contrib[i] = 0.0;
        if (tid < R_NBLK * 4) S.done[tid] = 0;
        if (tid < RNW) S.hcnt[tid] = 0;
        __syncthreads();
        const uint32_t item = uni(S.item);
        if (item >= n_items) break;
        Item it;
        if (FUSED) {
            it.q = item / fused_g;
            const uint32_t part = item - it.q * fused_g;
            it.doc_lo = (uint32_t)((unsigned long long)ix.n_docs * part / fused_g);
            it.doc_hi = (uint32_t)((unsigned long long)ix.n_docs * (part + 1) / fused_g);
            it.m = 0;  // the host sends only sparse queries of <= RT indexed terms this way
        } else {
            it = bt.items[item];
        }
        if (it.m > (uint32_t)RT) continue;  // more terms or dense (ITEM_DENSE): the other kernels'
        PROF_T(t_item);
        const uint32_t q = uni(it.q), lo = uni(it.doc_lo), hi = uni(it.doc_hi);
        uint32_t *hrow = bt.hist + (size_t)q * CUR_HB;

        auto theta_now = [&]() -> unsigned long long {
            const unsigned long long th = S.theta;
            return ((unsigned long long)uni((uint32_t)(th >> 32)) << 32) | uni((uint32_t)th);
        };
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

        // ---- a candidate document of the item: appended to the pool if it can still be among the hits.  The caller has
        // made sure of the room (S3: POOL_HIGH rule; cold pass: reservation).
        auto pool_push = [&](bool has, double sc, uint32_t d) {
            const unsigned long long sb = (unsigned long long)__double_as_longlong(sc);
            has = has && sb >= theta_now() && sb != 0ull;
            VCHK(!has || (sc * S.hscale < (double)CUR_HB && sc > 0.0), 21, d);
            if (has) {
                const uint32_t pos = atomicAdd(&S.pool_n, 1u);
                if (pos < POOL) {
                    S.pool_s[pos] = sb;
                    S.pool_d[pos] = d;
                } else {
                    S.fail = 4;  // (cannot happen: see the callers)
                }
                const double hb = sc * S.hscale;
                atomicAdd(&hrow[hb >= (double)(CUR_HB - 1) ? (uint32_t)(CUR_HB - 1) : (uint32_t)hb], 1u);
            }
        };

        // ---- tile planner (wave 0).  State in LDS between its turns.
        uint32_t p_cur = 0, p_end = 0, p_quota = 0, p_base = 0, p_st = NONE32, p_so = 0, p_df = 0, p_rank = 0, p_ne = 0, p_relax = 0;
        uint32_t p_cap = R_STAGE_SLOTS, p_tlo = lo, p_pf = 0;  // p_pf: the staged inputs belong to the cursors / quotas of now
        auto pl_load = [&]() {
            const bool act = lane < uni(S.mq);
            p_cur = S.pl[0][lane];
            p_quota = S.pl[1][lane];
            p_base = S.pl[2][lane];
            p_st = S.pl[3][lane];
            p_so = S.pl[4][lane];
            p_end = act ? S.t_b1[lane] : 0u;
            p_df = act ? S.t_df[lane] : 0u;
            p_rank = act ? (uint32_t)S.t_rank[lane] : 0u;
            p_ne = uni(S.plu[0]);
            p_relax = uni(S.plu[1]);
            p_cap = uni(S.plu[2]);
            p_tlo = uni(S.plu[3]);
            p_pf = uni(S.plu[4]);
        };
        auto pl_store = [&]() {
            S.pl[0][lane] = p_cur;
            S.pl[1][lane] = p_quota;
            S.pl[2][lane] = p_base;
            S.pl[3][lane] = p_st;
            S.pl[4][lane] = p_so;
            if (lane == 0) {
                S.plu[0] = p_ne;
                S.plu[1] = p_relax;
                S.plu[2] = p_cap;
                S.plu[3] = p_tlo;
                S.plu[4] = p_pf;
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
        auto plan_empty = [&](uint32_t buf) {
            if (lane == 0) S.hdr[buf] = make_uint4(p_tlo, p_tlo, 0, 0);
        };
        auto plan_tile = [&](uint32_t buf) {
            // MaxScore split (search.rs:153-169 is this test, one document at a time): the longest prefix of the
            // terms in ascending upper-bound order whose bounds sum below the threshold is NON-ESSENTIAL -- a
            // document made only of those terms cannot enter the top-k.  Their blocks are not planned at all.
            const uint32_t mqp = uni(S.mq);
            const double thd = __longlong_as_double((long long)theta_now());
            uint32_t p_th = 0;
            for (uint32_t pp = 1; pp <= mqp; ++pp)
                if (S.t_cum[pp] < thd) p_th = pp;
            if (p_th == mqp) {  // no document at all can reach the threshold any more
                plan_empty(buf);
                return;
            }
            // (p_relax: the essential lists intersect too densely even for the smallest tiles -- every term the
            // threshold allows becomes non-essential, whatever the lookups cost)
            const uint32_t p_new = !bt.ne_on ? 0u : p_relax ? p_th : (uint32_t)S.t_best[p_th];
            if (p_new > p_ne) {
                p_ne = p_new;
                assign_quotas();
                p_pf = 0;  // (the staged inputs were requested for the old quotas)
            }
            const double nesum = S.t_cum[p_ne];
            const bool alive = lane < mqp && p_rank >= p_ne && p_cur < p_end;
            if (!__ballot(alive) || p_tlo >= hi) {
                p_pf = 0;
                plan_empty(buf);
                return;
            }
            const uint32_t st = p_st < 64u ? p_st : 0u;
            const uint32_t cur_s = (uint32_t)__shfl((int)p_cur, (int)st), end_s = (uint32_t)__shfl((int)p_end, (int)st);
            const uint32_t quo_s = (uint32_t)__shfl((int)p_quota, (int)st);
            const uint32_t j = cur_s + p_so;
            const bool valid = p_st != NONE32 && p_so < quo_s && j < end_s;
            const bool want_bnd = alive && p_cur + p_quota < p_end;
            uint32_t bnd = NONE32;
            uint4 meta = make_uint4(NONE32, 0, 0, 0);
            double ub = 0.0;
            if (p_pf) {  // requested by LDS-DMA when the previous plan ended: no memory round trip here
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (want_bnd) bnd = S.pf_bnd[lane];
                if (valid) {
                    meta = S.pf_meta[lane];
                    ub = __hiloint2double((int)S.pf_ub[1][lane], (int)S.pf_ub[0][lane]);
                }
            } else {
                if (want_bnd) bnd = ix.blk_min_doc[p_cur + p_quota];
                if (valid) {
                    meta = ix.blk_meta[j];
                    ub = ix.blk_ub[j];
                }
            }
            uint32_t thi = min(hi, wave_min_u32(bnd));
            // a block that spans 2^16 documents or more is staged with 32-bit ids: two slots
            const bool wide = valid && meta.y - meta.x >= 65536u;
            // the 64 candidates (quotas) may hold more than a tile takes: the largest thi with <= p_cap slots
            auto slots_below = [&](uint32_t v) -> uint32_t {
                return (uint32_t)__popcll(__ballot(valid && meta.x < v)) + (uint32_t)__popcll(__ballot(wide && meta.x < v));
            };
            if (slots_below(thi) > p_cap) {
                uint32_t lo_v = p_tlo + 1, hi_v = thi;  // slots(lo_v) <= terms * 2 <= p_cap < slots(hi_v)
                while (hi_v - lo_v > 1) {
                    const uint32_t mid = lo_v + ((hi_v - lo_v) >> 1);
                    if (slots_below(mid) <= p_cap) lo_v = mid;
                    else hi_v = mid;
                }
                thi = lo_v;
            }
            const bool in_tile = valid && meta.x < thi;
            const unsigned long long mask = __ballot(in_tile);
            const unsigned long long wmask = __ballot(in_tile && wide);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const uint32_t wbefore = __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
            const uint32_t np = (uint32_t)__popcll(mask);
            if (in_tile) {
                const uint32_t md = (meta.w >> 8) & 0xff;
                const uint32_t fast = md <= 15u ? 1u : 0u;  // full bit-packed block whose id words fit one 256-byte DMA slot
                S.pm[buf][pos] = meta;
                S.pa[buf][pos] = make_uint2(j, p_st | (pos + wbefore) << 8 | (wide ? 1u : 0u) << 16 | fast << 17 | 1u << 18);
                S.pub[buf][pos] = __double2float_ru(ub);
            }
            if (lane >= np && lane < (uint32_t)R_NBLK) {  // unused entries decode nothing: width 0, ids far outside the tile
                S.pm[buf][lane] = make_uint4(0x7fffffffu, 0x7fffffffu, 0, 0);
                S.pa[buf][lane] = make_uint2(0, 0);
            }
            // No threshold yet (the query's first tile): every block's upper bound is the score of one of its postings
            // (the block's WAND pair, flush.rs:40-158), the blocks of ONE term hold distinct documents -- so the k-th largest
            // bound among a term's blocks of this tile is a lower bound of the query's k-th best score.
            double thd_c = thd;
            if (thd == 0.0 && ix.blk_ub_attained) {
                uint32_t best_t = 0, best_n = 0;
                for (uint32_t t = 0; t < mqp; ++t) {
                    const uint32_t nt = (uint32_t)__popcll(__ballot(in_tile && p_st == t));
                    if (nt > best_n) {
                        best_n = nt;
                        best_t = t;
                    }
                }
                if (best_n >= k) {
                    const bool member = in_tile && p_st == best_t;
                    unsigned long long mm = __ballot(member);
                    uint32_t rank = 0;  // members with a larger bound (ties: the lower lane first)
                    while (mm) {
                        const uint32_t jl = (uint32_t)__ffsll((long long)mm) - 1u;
                        mm &= mm - 1ull;
                        const double v = readlane_f64(ub, jl);
                        rank += (v > ub || (v == ub && jl < lane)) ? 1u : 0u;
                    }
                    const unsigned long long kth = __ballot(member && rank == k - 1u);
                    if (kth) {
                        const double t0 = readlane_f64(ub, (uint32_t)__ffsll((long long)kth) - 1u) * (1.0 - 4e-12);  // (the bound carries a factor 1 + 1e-12)
                        thd_c = t0;
                        if (lane == 0) {
                            atomicMax(&S.theta, (unsigned long long)__double_as_longlong(t0));
                            atomicMax(&bt.theta[q], (unsigned long long)__double_as_longlong(t0));
                        }
                    }
                }
            }
            // cold blocks (search.rs:203): upper bound at or above the threshold -- the threshold only rises, so
            // deciding here, one tile early, errs on the safe side.  Bit i of word w: entry (w - 1) + (RNW - 1) i
            if (lane < (uint32_t)RNW) S.coldw[buf][lane] = 0;
            const bool cold = in_tile && thd_c <= ub * (1.0 + 1e-7) + nesum;
            if (cold) atomicOr(&S.coldw[buf][1u + pos % (RNW - 1)], 1u << (pos / (RNW - 1)));
            const unsigned long long cmask = __ballot(in_tile && meta.y < thi);
            if (lane <= (uint32_t)RT) {  // lane t: entries before term t's slots = first entry of term t
                const unsigned long long below = p_base >= 64u ? ~0ull : ((1ull << p_base) - 1ull);
                S.ptb[buf][lane] = (uint8_t)__popcll(mask & below);
                const unsigned long long qm = p_quota >= 64u ? ~0ull : ((1ull << p_quota) - 1ull);
                if (p_base < 64u) p_cur += (uint32_t)__popcll((cmask >> p_base) & qm);
            }
            const unsigned long long cold_mask = __ballot(cold);
            const bool any_cold_blocks = cold_mask != 0ull;
            if (lane == 0) S.hdr[buf] = make_uint4(p_tlo, thi, np, p_ne | (any_cold_blocks ? 0x100u : 0u) | (uint32_t)__popcll(cold_mask) << 16);
            p_tlo = thi;
            {   // the next plan's inputs (same formulas on the advanced cursors), fetched by LDS-DMA while the tile is worked on
                const uint32_t cur_n = (uint32_t)__shfl((int)p_cur, (int)st);
                const uint32_t jn = cur_n + p_so;
                const bool valid_n = p_st != NONE32 && p_so < quo_s && jn < end_s;
                const bool alive_n = lane < mqp && p_rank >= p_ne && p_cur < p_end;
                if (valid_n) {
                    r_glds_dwordx4_v(&ix.blk_meta[jn], (uint32_t)(uintptr_t)&S.pf_meta[0]);
                    r_glds_dword_v(&ix.blk_ub[jn], (uint32_t)(uintptr_t)&S.pf_ub[0][0]);
                    r_glds_dword_v(reinterpret_cast<const uint32_t *>(&ix.blk_ub[jn]) + 1, (uint32_t)(uintptr_t)&S.pf_ub[1][0]);
                }
                if (alive_n && p_cur + p_quota < p_end) r_glds_dword_v(&ix.blk_min_doc[p_cur + p_quota], (uint32_t)(uintptr_t)&S.pf_bnd[0]);
                p_pf = 1;
            }
        };

        // ---- item setup (wave 0): terms, cursors, quotas, slot map; the first two plans
        if (wave == 0) {
            poll_request();
            uint32_t m = 0, term = NONE32;
            {
                const uint32_t qb = uni(bt.q_off[q]), qe = uni(bt.q_off[q + 1]);
                if (qe - qb <= 64) {  // one load per lane, compaction of the indexed terms through LDS
                    const uint32_t tt = lane < qe - qb ? bt.term_ids[qb + lane] : NONE32;
                    const bool ok = tt < ix.n_terms;  // search.rs:59-61
                    const unsigned long long okm = __ballot(ok);
                    if (ok) S.scratch[__builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = tt;
                    __builtin_amdgcn_wave_barrier();
                    m = (uint32_t)__popcll(okm);
                    if (lane < m) term = S.scratch[lane];
                    __builtin_amdgcn_wave_barrier();
                } else {
                    for (uint32_t p = qb; p < qe; ++p) {
                        const uint32_t tt = bt.term_ids[p];
                        if (tt >= ix.n_terms) continue;
                        if (m == lane) term = tt;
                        ++m;
                    }
                }
            }
            m = min(uni(m), (uint32_t)RT);  // (more than RT: filtered above through it.m; FUSED: the host checks)
            const bool act = lane < m;
            double s0 = 0.0, tub = 0.0;
            uint32_t df = 0;
            p_cur = p_end = 0;
            if (act) {
                const uint32_t b0 = ix.term_first_block[term], b1 = ix.term_first_block[term + 1];
                s0 = ix.term_s0[term];
                df = ix.term_df[term];
                const double wtf = (double)ix.term_wand_tf[term];
                tub = ((wtf * s0) / (wtf + S.s1[ix.term_wand_fn[term]])) * (1.0 + 1e-12);
                p_cur = r_first_block_ge(ix, b0, b1, lo);
                p_end = b1;
                S.t_s0[lane] = s0;
                S.t_ub[lane] = tub;
                S.t_b0[lane] = b0;
                S.t_b1[lane] = b1;
                S.t_df[lane] = df;
            }
            // terms in ascending order of their token upper bound; prefix sums; admissible prefixes
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
                // a prefix of p terms is admissible when its shortest list is still ne_ratio times longer
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
                    if (pp + 1 < m && (dfo >= (unsigned long long)bt.ne_ratio * (sumdf - head) || dfo * 16ull >= ix.n_docs)) best = pp + 1;
                    if (lane == 0) {
                        S.t_cum[pp + 1] = cum;
                        S.t_best[pp + 1] = (uint8_t)best;
                        S.t_ord[pp] = (uint8_t)owner;
                    }
                }
            }
            p_ne = 0;
            p_relax = 0;
            p_cap = R_STAGE_SLOTS;
            if (lane == 0) S.mq = m;
            __builtin_amdgcn_wave_barrier();
            assign_quotas();
            const double hscale = (double)CUR_HB / sums0;  // score -> histogram bucket: linear in [0, sum of s0)
            if (lane == 0) {
                if constexpr (FUSED) {
                    // the records plan_kernel would have made: scan_many_kernel (items this kernel gives up) and merge_kernel
                    // (queries with such an item) of the general route read them
                    Item rec;
                    rec.q = q;
                    rec.doc_lo = lo;
                    rec.doc_hi = hi;
                    rec.m = m;
                    bt.items[item] = rec;
                    if (item == 0) *bt.n_items = n_items;
                    if (item % fused_g == 0) bt.q_item_base[q] = item;
                    if (item + 1 == n_items) bt.q_item_base[q + 1] = n_items;
                }
                S.q = q;
                S.lo = lo;
                S.hi = hi;
                S.hscale = hscale;
            }
            __builtin_amdgcn_wave_barrier();
            poll_consume();
            p_tlo = lo;
            if (m == 0) plan_empty(0);
            else plan_tile(0);
            pl_store();
        }
        __syncthreads();
        const uint32_t mq = uni(S.mq);

        // ---- LDS-DMA of this wave's blocks of the tile planned in buf: 256 bytes of id words per entry into the entry's
        // raw slot (a full block of width <= 15 reads at most 16 * 15 + 16 + 8 bytes; the rest of the slot is slack)
        auto dma_issue = [&](uint32_t buf) {
            const uint32_t np1 = uni(S.hdr[buf].z);
            uint32_t off8[RB], fl[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {  // (all the descriptor reads in flight together)
                const uint32_t e = (wave - 1u) + (RNW - 1) * i;
                off8[i] = S.pm[buf][e].z;
                fl[i] = S.pa[buf][e].y;
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const uint32_t e = (wave - 1u) + (RNW - 1) * i;
                if (e < np1 && ((uni(fl[i]) >> 17) & 1u)) {
                    const uint32_t o8 = uni(off8[i]);
                    VCHK(8ull * o8 < ix.blob_bytes, 32, o8);
#ifdef R_NO_DMA  // (tools: plain loads + LDS stores instead of the LDS-DMA)
                    reinterpret_cast<uint32_t *>(&S.raw[e * 16])[lane] = *reinterpret_cast<const uint32_t *>(ix.blob + 8ull * o8 + 4u * lane);
#else
                    r_glds_dword(ix.blob + 8ull * o8, 4u * lane, (uint32_t)(uintptr_t)&S.raw[e * 16]);
#endif
                }
            }
        };

        // ---- exact score of documents by lookups: a candidate per lane (all 64 lanes call; `cand` marks the lanes that hold
        // one and is cleared for those that cannot reach the threshold).  em: the terms whose contribution is known already --
        // from the row (row != NONE32) or the single posting (tself, pself); the other terms are looked up: block upper
        // bounds first (search.rs:177-203), then the block is decoded by the wave and the candidate's posting found by
        // comparison.  Sum in ascending key order (evaluate.rs:43-72).
        auto complete = [&](bool &cand, uint32_t d, uint32_t row, uint32_t tself, double pself, double partial, uint32_t em, double nes) -> double {
            const double thd = __longlong_as_double((long long)theta_now());
            cand = cand && partial + nes >= thd;
            if (!__ballot(cand)) return 0.0;
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
            if (!__ballot(cand)) return 0.0;
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
                    for (;;) {  // one candidate at a time: its block decoded by the wave, the posting found by comparison
                        const unsigned long long pmask = __ballot(pend);
                        if (!pmask) break;
                        const uint32_t src = (uint32_t)__ffsll((long long)pmask) - 1u;
                        const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)src);
                        const uint32_t dq = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)src);
                        const uint4 bm = uni4(ix.blk_meta[blk]);
                        const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
                        uint32_t a0, a1;
                        decode_doc_ids(ix.blob + 8ull * bm.z, md, n, bm.x, lane, a0, a1);
                        const bool h0 = 2 * lane < n && a0 == dq, h1 = 2 * lane + 1 < n && a1 == dq;
                        const unsigned long long hm = __ballot(h0 || h1);
                        double cq = 0.0;
                        if (hm) {  // the lane that holds the posting fetches its tf / fieldnorm
                            const uint32_t hl = (uint32_t)__ffsll((long long)hm) - 1u;
                            double cv = 0.0;
                            if (lane == hl) {
                                const uint32_t idx = 2 * lane + (h0 ? 0u : 1u);
                                const uint8_t *tbody = ix.blob + 8ull * bm.z + ((payload_bytes(md, n) + 7u) & ~7u);
                                const FieldAddr fa = field_addr(mt, n, idx);
                                const uint32_t flo = *reinterpret_cast<const uint32_t *>(tbody + fa.off0);
                                const uint32_t fhi = *reinterpret_cast<const uint32_t *>(tbody + fa.off1);
                                const uint32_t fn = ix.post_fn[128ull * blk + idx];
                                const double tf = (double)field_val(flo, fhi, fa);
                                cv = (tf * S.t_s0[t]) / (tf + S.s1[fn]);  // Cache::evaluate, bm25.rs:355-358
                            }
                            cq = readlane_f64(cv, hl);
                        }
                        if (lane == src) {
                            c = cq;
                            pend = false;
                        }
                    }
                }
                acc += c;
            }
            return acc;
        };

        // ---- pool selection: raise the threshold to (a lower bound of) the k-th best score of the pool and drop what is
        // below it.  All eight waves; barriers inside.  Up to three levels of 256 buckets: linear in the score, each level
        // refining the bucket that holds the k-th entry.
        uint32_t tile_dbg = 0;
        auto shrink_pool = [&]() {
            const uint32_t n = min(uni(S.pool_n), POOL);
            uint32_t *bh = S.bm;  // the seen filter is clean and idle wherever the pool is shrunk: its first 256 words are the histogram
            if (tid == 0) {
                S.sel_lo = 0.0;
                S.sel_mul = 1.0;
                S.sel_need = k;
                S.sel_stop = n < k ? 1u : 0u;  // fewer than k entries: nothing to select
            }
            lds_barrier();
            for (int level = 0; level < 3 && !uni(S.sel_stop); ++level) {
                if (tid < CUR_HB) bh[tid] = 0;
                lds_barrier();
                const double flo = S.sel_lo, fmul = S.sel_mul, hs = S.hscale;
                for (uint32_t i = tid; i < n; i += RWG) {
                    const double v = (__longlong_as_double((long long)S.pool_s[i]) * hs - flo) * fmul;
                    if (v >= 0.0 && v < (double)CUR_HB) atomicAdd(&bh[(uint32_t)v], 1u);
                    // v >= 256: counted above the focus by an earlier level (level 0: clamped below)
                    else if (level == 0 && v >= (double)CUR_HB) atomicAdd(&bh[CUR_HB - 1], 1u);
                }
                lds_barrier();
                if (wave == 0) {  // the highest bucket b with `need` entries at or above it
                    const uint4 c4 = *reinterpret_cast<const uint4 *>(&bh[4 * lane]);
                    const uint32_t need = uni(S.sel_need);
                    const uint32_t own = c4.x + c4.y + c4.z + c4.w;
                    const uint32_t incl = wave_incl_scan_u32(own);
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    const uint32_t above = total - incl;
                    const unsigned long long hit = __ballot(above + own >= need);
                    if (hit) {
                        const uint32_t hl = 63u - (uint32_t)__builtin_clzll(hit);
                        uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)above, (int)hl), b = 4 * hl;
                        const uint32_t c3 = (uint32_t)__builtin_amdgcn_readlane((int)c4.w, (int)hl);
                        const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)c4.z, (int)hl);
                        const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)c4.y, (int)hl);
                        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c4.x, (int)hl);
                        uint32_t inb = c0;
                        if (a + c3 >= need) {
                            b += 3;
                            inb = c3;
                        } else if (a + c3 + c2 >= need) {
                            b += 2;
                            a += c3;
                            inb = c2;
                        } else if (a + c3 + c2 + c1 >= need) {
                            b += 1;
                            a += c3 + c2;
                            inb = c1;
                        } else {
                            a += c3 + c2 + c1;
                        }
                        if (lane == 0) {
                            S.sel_lo = S.sel_lo + (double)b / S.sel_mul;
                            S.sel_mul = S.sel_mul * (double)CUR_HB;
                            S.sel_need = need - a;       // entries still to be found inside bucket b
                            // kept = the entries above the bucket + the bucket: small enough -> stop refining
                            if ((k - (need - a)) + inb <= POOL_HIGH / 2u + (uint32_t)KMAX / 2u) S.sel_stop = 1;
                        }
                    } else if (lane == 0) {
                        S.sel_stop = 1;  // (fewer than `need` entries in the focus: keep the focus edge)
                    }
                }
                lds_barrier();
            }
            // threshold = lower edge of the focus (every counted entry has score * hscale >= it up to roundings far
            // below the margin), compaction of the entries at or above it
            const double edge = (S.sel_lo / S.hscale) * (1.0 - 1e-12);
            const unsigned long long eb = n < k ? 0ull : (unsigned long long)__double_as_longlong(edge > 0.0 ? edge : 0.0);
            VCHK(edge * S.hscale < (double)CUR_HB, 29, n);
            unsigned long long ks[(POOL + RWG - 1) / RWG];
            uint32_t kd[(POOL + RWG - 1) / RWG];
#pragma unroll
            for (int j = 0; j < (int)((POOL + RWG - 1) / RWG); ++j) {
                const uint32_t i = tid + j * RWG;
                ks[j] = i < n ? S.pool_s[i] : 0ull;
                kd[j] = i < n ? S.pool_d[i] : 0u;
#ifdef VBM25_CHECK
                if (i < n && !(__longlong_as_double((long long)ks[j]) * S.hscale < (double)CUR_HB) && atomicCAS(&bt.dbg[0], 0u, 36u) == 0u) {
                    bt.dbg[1] = i;
                    bt.dbg[2] = S.item;
                    bt.dbg[3] = n;
                    bt.dbg[4] = (uint32_t)ks[j];
                    bt.dbg[5] = (uint32_t)(ks[j] >> 32);
                    bt.dbg[6] = kd[j];
                    bt.dbg[7] = tile_dbg;
                    bt.dbg[8] = (uint32_t)S.pool_s[i > 0 ? i - 1 : 0];
                    bt.dbg[9] = (uint32_t)(S.pool_s[i > 0 ? i - 1 : 0] >> 32);
                    bt.dbg[10] = S.pool_d[i > 0 ? i - 1 : 0];
                    bt.dbg[11] = (uint32_t)S.pool_s[i + 1 < n ? i + 1 : i];
                    bt.dbg[12] = (uint32_t)(S.pool_s[i + 1 < n ? i + 1 : i] >> 32);
                    bt.dbg[13] = S.pool_d[i + 1 < n ? i + 1 : i];
                    bt.dbg[14] = (uint32_t)__double_as_longlong(S.hscale);
                    bt.dbg[15] = (uint32_t)(__double_as_longlong(S.hscale) >> 32);
                }
#endif
            }
            if (tid == 0) {
                S.pool_w = 0;
                atomicMax(&S.theta, eb);
                if (eb) atomicMax(&bt.theta[q], eb);
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < (int)((POOL + RWG - 1) / RWG); ++j) {
                const uint32_t i = tid + j * RWG;
                if (i < n && ks[j] >= eb) {
                    const uint32_t pos = atomicAdd(&S.pool_w, 1u);
                    S.pool_s[pos] = ks[j];
                    S.pool_d[pos] = kd[j];
                }
            }
            lds_barrier();
            if (tid < CUR_HB) bh[tid] = 0;  // (the filter's words again)
            if (tid == 0) {
                S.pool_n = S.pool_w;
                if (S.pool_w > POOL_HIGH) S.fail = 5;  // masses of equal scores: the exhaustive kernel's case
            }
            lds_barrier();
#ifdef VBM25_PROFILE
            prof[14] += 1;
#endif
        };

        // ---- late list: documents that found no row / hit slot, scored by lookups in every list.  All waves.
        auto flush_late = [&]() {
            const uint32_t n = min(uni(S.nlate), (uint32_t)R_LATE);
            for (uint32_t base = wave * 64u; base < n; base += RWG) {
                const bool has = base + lane < n;
                const uint32_t d = has ? S.late[base + lane] : 0u;
                bool cand = has;
                const double acc = complete(cand, d, NONE32, NONE32, 0.0, 0.0, 0u, 1e300);
                VCHK(!cand || acc * S.hscale < (double)CUR_HB, 37, d);
                pool_push(cand, acc, d);  // (at most R_LATE / 2 + a tile's extra rows at a time: see the room rule below)
            }
            lds_barrier();
            if (tid == 0) S.nlate = 0;
            lds_barrier();
        };
        auto late_push = [&](uint32_t d) {
            const uint32_t pos = atomicAdd(&S.nlate, 1u);
            if (pos < (uint32_t)R_LATE) S.late[pos] = d;
            else S.fail = 3;
        };

        // =====================================================================
        // Tile loop
        // =====================================================================
        bool failed = false;
        PROF_T(t_loop);
        PROF_ADD(8, t_item, t_loop);
        if (wave != 0) dma_issue(0);
        const uint32_t lane_raw = 8u * (lane & 1u);          // byte offset of the lane's stream pair inside a 16-byte group
        const uint32_t lane_half = lane >> 1;
        uint32_t np_prev = 0;

        for (uint32_t tile = 0;; ++tile) {
            tile_dbg = tile;
            const uint32_t buf = tile % R_PLAN_RING, pbuf = (tile + R_PLAN_RING - 1) % R_PLAN_RING;
            const uint32_t par = tile & 1u;
            const uint4 hdr = uni4(S.hdr[buf]);
            const uint32_t tlo = hdr.x, thi = hdr.y, np = hdr.z;
            if (np == 0 && np_prev == 0) break;
            const uint32_t span = thi - tlo;
            PROF_T(t_a);

            // ---- hits of the previous tile: tf / fieldnorm of every hit of this wave in one batch of loads; the values
            // are consumed after S1 / the plan
            const uint32_t nh = min(uni(S.hcnt[wave]), (uint32_t)R_HITS);
            uint32_t h_lo = 0, h_hi = 0, h_fn = 0, h_sh = 0, h_mask = 0, h_rt = NONE32;
            auto hits_issue = [&](uint32_t base) {
                h_rt = NONE32;
                if (base + lane < nh) {
                    const uint32_t rec = S.hits[wave][base + lane];
                    const uint32_t r = rec >> 16, e = (rec >> 8) & 0xffu, idx = rec & 0xffu;
                    VCHK(e < (uint32_t)R_NBLK && idx < 128u && r < ROWS, 23, rec);
                    const uint4 sj = S.pm[pbuf][e < (uint32_t)R_NBLK ? e : 0u];
                    const uint2 aux = S.pa[pbuf][e < (uint32_t)R_NBLK ? e : 0u];
                    VCHK(aux.x < ix.n_blocks && (aux.y & 0xffu) < mq && ((aux.y >> 18) & 1u), 24, aux.y);
                    VCHK(8ull * sj.z < ix.blob_bytes, 30, sj.z);
#ifdef VBM25_CHECK
                    if (!(e < (uint32_t)R_NBLK && idx < 128u && r < ROWS && aux.x < ix.n_blocks && (aux.y & 0xffu) < mq && 8ull * sj.z < ix.blob_bytes)) return;
#endif
                    const uint32_t nj = sj.w & 0xff, mdj = (sj.w >> 8) & 0xff, mtj = (sj.w >> 16) & 0xff;
                    const uint8_t *tbody = ix.blob + 8ull * sj.z + ((payload_bytes(mdj, nj) + 7u) & ~7u);
                    const FieldAddr fa = field_addr(mtj, nj, idx);
                    h_lo = *reinterpret_cast<const uint32_t *>(tbody + fa.off0);
                    h_hi = *reinterpret_cast<const uint32_t *>(tbody + fa.off1);
                    h_fn = ix.post_fn[128ull * aux.x + idx];
                    h_sh = fa.sh;
                    h_mask = fa.mask;
                    h_rt = r << 8 | (aux.y & 0xffu);
                }
            };
            auto hits_consume = [&]() {
                if (h_rt != NONE32) {
                    const uint32_t r = h_rt >> 8, t = h_rt & 0xffu;
                    const double tf = (double)(__builtin_amdgcn_alignbit(h_hi, h_lo, h_sh) & h_mask);
                    const double cv = (tf * S.t_s0[t]) / (tf + S.s1[h_fn]);  // Cache::evaluate, bm25.rs:355-358
                    VCHK(cv * S.hscale < (double)CUR_HB && cv > 0.0 && t < mq && r < ROWS, 26, h_rt);
                    S.contrib[(r << LRT) + t] = cv;
                }
            };

            // ---- the previous tile's tail: barrier B, pool housekeeping, cold pass
            if (tile != 0) {
                lds_barrier();  // ---- B: done bits and hit records of the previous tile complete; filter and hash set clean
                PROF_T(t_f);
                PROF_ADD(4, t_a, t_f);
                if (uni(S.fail)) {
                    failed = true;
                    break;
                }
                bool stop = false;
                {
                    // (the previous tile's plan, by the names the cold pass uses)
                    const uint4 phdr = uni4(S.hdr[pbuf]);
                    const uint32_t buf = pbuf, tlo = phdr.x, span = phdr.y - phdr.x, np = phdr.z, pne = phdr.w & 0xffu;
                    const bool any_cold = (phdr.w & 0x100u) != 0;
                    const uint32_t n_cold = (phdr.w >> 16) & 0xffu;
                    uint32_t nv = wave != 0 ? (np + (RNW - 1) - wave) / (RNW - 1) : 0u;
                    if (nv > (uint32_t)RB) nv = RB;
                    (void)np;
                    uint32_t pending = 0;
                    if (wave != 0 && any_cold) pending = uni(S.coldw[buf][wave]) & ((1u << nv) - 1u);
                    const bool boot = any_cold && uni(S.theta_zero) != 0u;  // (snapshot taken before barrier B: uniform)
                    const double nesum = pne ? S.t_cum[pne] : 0.0;
                    uint32_t emask = 0xffffffffu;  // bit t: term t is essential
                    if (pne && any_cold) {
                        emask = 0;
                        for (uint32_t t = 0; t < mq; ++t) emask |= ((uint32_t)S.t_rank[t] >= pne ? 1u : 0u) << t;
                    }
                    // mode 0: push (a block that finds no room stays pending); mode 1: histogram of the scores only; mode 2: push,
                    // postings that find no room go to the late list
                    auto cold_blocks = [&](uint32_t mode) {
                        uint32_t *bh = S.bm;
                        // (the bootstrap histogram takes a sample: the wave's first cold block -- the k-th best of any set of
                        // distinct documents is a lower bound of the k-th best of all)
                        uint32_t todo = mode == 1 ? pending & (0u - pending) : pending;
                        while (todo) {
                            uint32_t gi[4];
                            uint32_t n4 = 0;
        #pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                gi[g] = 0;
                                if (todo) {
                                    gi[g] = (uint32_t)__ffs((int)todo) - 1u;
                                    todo &= todo - 1u;
                                    ++n4;
                                }
                            }
                            uint32_t l0[4], h0[4], l1[4], h1[4], fnp[4];
                            uint32_t skip = 0;  // bit g: block not scored in this pass (below the threshold: done; no room: next round)
                            const double thd = __longlong_as_double((long long)theta_now());
        #pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                l0[g] = h0[g] = l1[g] = h1[g] = fnp[g] = 0;
                                if ((uint32_t)g < n4) {
                                    const uint32_t e = (wave - 1u) + (RNW - 1) * gi[g];
                                    const float pubv = __uint_as_float(uni(__float_as_uint(S.pub[buf][e])));
                                    if (thd > (double)pubv * (1.0 + 1e-7) + nesum) {  // the planner decided one tile early: the threshold of now
                                        pending &= ~(1u << gi[g]);
                                        skip |= 1u << g;
                                    } else if (mode == 0 && uni(S.pool_n) >= POOL_COLD) {
                                        skip |= 1u << g;  // no room at all: next round
                                    } else {
                                        const uint4 sj = uni4(S.pm[buf][e]);
                                        const uint32_t blkj = uni(S.pa[buf][e].x);
                                        const uint32_t nj = sj.w & 0xff, mdj = (sj.w >> 8) & 0xff, mtj = (sj.w >> 16) & 0xff;
                                        VCHK(e < np && blkj < ix.n_blocks && 8ull * sj.z < ix.blob_bytes, 31, e);
                                        const uint8_t *tbody = ix.blob + 8ull * sj.z + ((payload_bytes(mdj, nj) + 7u) & ~7u);
                                        const FieldAddr f0 = field_addr(mtj, nj, 2 * lane), f1 = field_addr(mtj, nj, 2 * lane + 1);
                                        l0[g] = *reinterpret_cast<const uint32_t *>(tbody + f0.off0);
                                        h0[g] = *reinterpret_cast<const uint32_t *>(tbody + f0.off1);
                                        l1[g] = *reinterpret_cast<const uint32_t *>(tbody + f1.off0);
                                        h1[g] = *reinterpret_cast<const uint32_t *>(tbody + f1.off1);
                                        fnp[g] = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * blkj)[lane];
                                    }
                                }
                            }
        #pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                if ((uint32_t)g < n4 && !((skip >> g) & 1u)) {
                                    const uint32_t i = gi[g];
                                    const uint32_t e = (wave - 1u) + (RNW - 1) * i;
                                    const uint4 sj = uni4(S.pm[buf][e]);
                                    const uint32_t fl = uni(S.pa[buf][e].y), t = fl & 0xffu, slot = (fl >> 8) & 0xffu;
                                    uint32_t rel0, rel1;
                                    if ((fl >> 16) & 1u) {
                                        const uint2 dd = *reinterpret_cast<const uint2 *>(&S.stage[slot * (uint32_t)R_SS + 2u * lane]);
                                        rel0 = dd.x;
                                        rel1 = dd.y;
                                    } else {
                                        const uint32_t dd = S.stage[slot * (uint32_t)R_SS + lane];
                                        rel0 = dd & 0xffffu;
                                        rel1 = dd >> 16;
                                    }
                                    const uint32_t nj = sj.w & 0xff, mtj = (sj.w >> 16) & 0xff;
                                    const uint32_t d0 = sj.x + rel0, d1 = sj.x + rel1;
                                    const uint32_t dwi = S.done[e * 4 + (lane >> 4)];
                                    bool ok0 = 2 * lane < nj && d0 - tlo < span && !((dwi >> ((2 * lane) & 31)) & 1u);
                                    bool ok1 = 2 * lane + 1 < nj && d1 - tlo < span && !((dwi >> ((2 * lane + 1) & 31)) & 1u);
                                    const FieldAddr f0 = field_addr(mtj, nj, 2 * lane), f1 = field_addr(mtj, nj, 2 * lane + 1);
                                    const double s0t = S.t_s0[t];
                                    const double tf0 = (double)field_val(l0[g], h0[g], f0), tf1 = (double)field_val(l1[g], h1[g], f1);
                                    double p0 = (tf0 * s0t) / (tf0 + S.s1[fnp[g] & 0xff]);
                                    double p1 = (tf1 * s0t) / (tf1 + S.s1[fnp[g] >> 8]);
                                    if (mode == 1) {  // (non-essential terms: the single-term score is a lower bound of the document's -- as good)
                                        if (ok0) atomicAdd(&bh[min((uint32_t)(p0 * S.hscale), (uint32_t)(CUR_HB - 1))], 1u);
                                        if (ok1) atomicAdd(&bh[min((uint32_t)(p1 * S.hscale), (uint32_t)(CUR_HB - 1))], 1u);
                                        continue;
                                    }
                                    if (pne != 0 && __ballot(ok0 || ok1)) {  // completion by lookups in the non-essential lists
#pragma nounroll
                                        for (uint32_t si = 0; si < 2; ++si) {  // (one call site for the lane's two postings)
                                            bool okx = si ? ok1 : ok0;
                                            const double px = si ? p1 : p0;
                                            const double cx = complete(okx, si ? d1 : d0, NONE32, t, px, px, emask, nesum);
                                            if (si) {
                                                ok1 = okx;
                                                p1 = cx;
                                            } else {
                                                ok0 = okx;
                                                p0 = cx;
                                            }
                                        }
                                    }
                                    const unsigned long long thb = theta_now();
                                    const bool a0 = ok0 && (unsigned long long)__double_as_longlong(p0) >= thb && p0 != 0.0;
                                    const bool a1 = ok1 && (unsigned long long)__double_as_longlong(p1) >= thb && p1 != 0.0;
                                    const unsigned long long am0 = __ballot(a0), am1 = __ballot(a1);
                                    const uint32_t c0 = (uint32_t)__popcll(am0), cnt = c0 + (uint32_t)__popcll(am1);
                                    if (cnt) {  // room for exactly the postings that pass, reserved with one atomic
                                        // (compare-and-swap, not add-then-undo: an undo that is not the last reservation leaves a hole)
                                        uint32_t base = NONE32;
                                        if (lane == 0) {
                                            uint32_t old = S.pool_n;
                                            while (old + cnt <= POOL_COLD) {
                                                const uint32_t prev = atomicCAS(&S.pool_n, old, old + cnt);
                                                if (prev == old) {
                                                    base = old;
                                                    break;
                                                }
                                                old = prev;
                                            }
                                        }
                                        base = uni(base);
                                        if (base == NONE32) {
                                            if (mode != 2) continue;  // next round (after the shrink)
                                            if (a0) late_push(d0);    // no rounds on this path: the lookup path scores them
                                            if (a1) late_push(d1);
                                            pending &= ~(1u << i);
                                            continue;
                                        }
                                        const uint32_t q0 = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(am0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am0, 0u));
                                        const uint32_t q1 = base + c0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(am1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am1, 0u));
                                        VCHK(!a0 || p0 * S.hscale < (double)CUR_HB, 22, d0);
                                        VCHK(!a1 || p1 * S.hscale < (double)CUR_HB, 22, d1);
                                        if (a0) {
                                            S.pool_s[q0] = (unsigned long long)__double_as_longlong(p0);
                                            S.pool_d[q0] = d0;
                                            const double hb = p0 * S.hscale;
                                            atomicAdd(&hrow[hb >= (double)(CUR_HB - 1) ? (uint32_t)(CUR_HB - 1) : (uint32_t)hb], 1u);
                                        }
                                        if (a1) {
                                            S.pool_s[q1] = (unsigned long long)__double_as_longlong(p1);
                                            S.pool_d[q1] = d1;
                                            const double hb = p1 * S.hscale;
                                            atomicAdd(&hrow[hb >= (double)(CUR_HB - 1) ? (uint32_t)(CUR_HB - 1) : (uint32_t)hb], 1u);
                                        }
                                    }
                                    pending &= ~(1u << i);
        #ifdef VBM25_PROFILE
                                    prof[11] += 1;
        #endif
                                }
                            }
                        }
                    };
                    // `sure`: every cold block of the tile fits the pool whatever passes -- no retry rounds, no barrier: the waves
                    // without cold blocks go on to their next tile while the others score theirs
                    const bool sure = !boot && uni(S.pool_snap) <= POOL_HIGH && uni(S.late_snap) <= (uint32_t)R_LATE / 2u &&
                                      uni(S.pool_snap) + 128u * n_cold <= POOL_COLD;
                    uint32_t mode = boot ? 1u : 0u;  // 1: no threshold yet -- histogram of the single-term scores first
#ifdef VBM25_PROFILE
                    if (wave != 0 && !sure) prof[10] += 1;
#endif
                    for (uint32_t round = 0; !sure || any_cold; ++round) {
                        if (!sure && mode == 0) {
                            // (uniform decisions: the snapshots were taken before barrier B, a later round follows barrier C)
                            const bool late_full = round == 0 ? uni(S.late_snap) > (uint32_t)R_LATE / 2u : uni(S.nlate) > (uint32_t)R_LATE / 2u;
                            const bool pool_full = round == 0 ? uni(S.pool_snap) > POOL_HIGH : true;
                            if (pool_full || late_full) {
                                shrink_pool();
                                // (the late documents' scores fit the shrunk pool; the next tile's check sees them)
                                if (late_full && !uni(S.fail)) flush_late();
                            }
                            if (uni(S.fail)) {
                                stop = true;
                                break;
                            }
                        }
                        if (!any_cold) break;
                        cold_blocks(sure ? 2u : mode);
                        if (sure) break;
                        if (mode == 1) {
                            // the k-th best single-term score of the tile's cold postings (documents with one posting in the
                            // tile: distinct, and the score is the document's) starts the threshold
                            lds_barrier();
                            if (wave == 0) {
                                const uint4 c4 = *reinterpret_cast<const uint4 *>(&S.bm[4 * lane]);
                                const uint32_t own = c4.x + c4.y + c4.z + c4.w;
                                const uint32_t incl = wave_incl_scan_u32(own);
                                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                                const uint32_t above = total - incl;
                                const unsigned long long hit = __ballot(above + own >= k);
                                if (hit) {
                                    const uint32_t hl = 63u - (uint32_t)__builtin_clzll(hit);
                                    uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)above, (int)hl), b = 4 * hl;
                                    const uint32_t c3 = (uint32_t)__builtin_amdgcn_readlane((int)c4.w, (int)hl);
                                    const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)c4.z, (int)hl);
                                    const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)c4.y, (int)hl);
                                    if (a + c3 >= k) b += 3;
                                    else if (a + c3 + c2 >= k) b += 2;
                                    else if (a + c3 + c2 + c1 >= k) b += 1;
                                    const double edge = ((double)b / S.hscale) * (1.0 - 1e-12);
                                    if (lane == 0) atomicMax(&S.theta, (unsigned long long)__double_as_longlong(edge));
                                }
                            }
                            lds_barrier();
                            if (tid < CUR_HB) S.bm[tid] = 0;  // (the filter's words again; barrier C of the push pass orders this before the next S1)
                            mode = 0;
                            --round;  // (the push pass is this round's)
                            continue;
                        }
                        if (pending && lane == 0) S.cold_retry = 1;
                        lds_barrier();  // ---- C (tiles whose cold blocks may not fit the pool at once)
                        const bool again = uni(S.cold_retry) != 0;
                        lds_barrier();
                        if (!again) break;
                        if (tid == 0) S.cold_retry = 0;
                        if (round >= 64u) {  // (a block that never fits: masses of equal scores)
                            if (tid == 0) S.fail = 6;
                            lds_barrier();
                            stop = true;
                            break;
                        }
                    }
                    // this wave's done bits of that tile
                    if (wave != 0 && lane < 4 * RB) S.done[((wave - 1u) + (RNW - 1) * (lane >> 2)) * 4 + (lane & 3)] = 0;
                }
                if (stop) {
                    failed = true;
                    break;
                }
                PROF_T(t_g);
                PROF_ADD(5, t_f, t_g);
#ifdef VBM25_PROFILE
                if (wave != 0 && tile == 1) prof[8] += t_g - t_f;  // (workers: the first tile's tail = the threshold bootstrap)
#endif
            }
            // ---- S1a (workers): decode into registers.  The raw words of tile `tile` were requested one tile ago.
            PROF_T(t_a0);
            uint32_t nv = 0;  // this wave's entries: (wave - 1) + 7 i < np  <=>  i < nv
            uint32_t r0[RB], r1[RB];     // ids relative to the block's first id, two per lane
            uint32_t bmin[RB], bfl[RB];  // per entry (uniform values kept in VGPRs): min_doc - tlo; slot << 8 | wide << 16 | fast << 17 | width << 24
            bool allfast = true;
            if (wave != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                hits_issue(0);
                for (uint32_t base = 64; base < nh; base += 64) {  // (every record read before S1 re-uses the list for its events)
                    hits_consume();
                    hits_issue(base);
                }
                nv = (np + (RNW - 1) - wave) / (RNW - 1);
                if (nv > (uint32_t)RB) nv = RB;
                asm volatile("; MARK_S1A_BEGIN");
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const uint32_t e = (wave - 1u) + (RNW - 1) * i;
                    const uint4 c = S.pm[buf][e];
                    const uint2 a = S.pa[buf][e];
                    const uint32_t fast = (a.y >> 17) & 1u;
                    bmin[i] = c.x - tlo;
                    bfl[i] = (a.y & 0x00ffff00u) | (fast ? (c.w >> 8) & 0xffu : 0u) << 24;  // (a block off the fast path decodes as width 0 here)
                    allfast = allfast && ((uint32_t)i >= nv || fast != 0u);
                }
                // full bit-packed blocks (compression.rs:65-92), branch-free so that the decodes overlap; unused entries have width 0
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const uint32_t e = (wave - 1u) + (RNW - 1) * i;
                    const uint32_t width = bfl[i] >> 24;
                    const uint32_t bit = __umul24(lane_half, width);
                    const uint8_t *rp = reinterpret_cast<const uint8_t *>(&S.raw[e * 16]) + ((bit >> 1) & ~15u) + lane_raw;
                    const uint2 wa = *reinterpret_cast<const uint2 *>(rp);
                    const uint2 wb = *reinterpret_cast<const uint2 *>(rp + 16);
                    const uint32_t mask = (1u << width) - 1u;  // width <= 15
                    const uint32_t v0 = __builtin_amdgcn_alignbit(wb.x, wa.x, bit) & mask;  // ((hi:lo) >> (bit mod 32))
                    const uint32_t v1 = __builtin_amdgcn_alignbit(wb.y, wa.y, bit) & mask;
                    const uint32_t own = v0 + v1;
                    const uint32_t incl = wave_incl_scan_u32(own);
                    r0[i] = (incl - own) + v0;
                    r1[i] = r0[i] + v1;
                }
                asm volatile("; MARK_S1A_END");
            }
            PROF_T(t_a1);

            bool dma_done = false;
            if (wave == 0) {
                // ---- control wave: plan of the next tile (read by the others after barrier A), hits, threshold poll
                PROF_T(t_p0);
                pl_load();
                {   // rows run short: smaller tiles (the planner's cap follows the rows the last tiles needed)
                    const uint32_t seen = uni(S.rows_seen);
                    if (seen > ROWS * 3u / 4u) p_cap = max(p_cap >> 1, 2u * mq);
                    else if (seen < ROWS / 4u) p_cap = min(p_cap + (p_cap >> 2) + 1u, (uint32_t)R_STAGE_SLOTS);
                }
                plan_tile((tile + 1) % R_PLAN_RING);
                pl_store();
                if (lane == 0) S.plan_seq = tile + 1u;  // (the workers may request the next tile's id words before barrier A)
                PROF_T(t_p1);
                PROF_ADD(7, t_p0, t_p1);
                if ((tile & 3u) == 3u) poll_request();  // the query's shared threshold (other items'), every fourth tile
                for (uint32_t base = 0; base < nh; base += 64) {
                    hits_issue(base);
                    hits_consume();
                }
                if ((tile & 3u) == 3u) poll_consume();
                PROF_T(t_p);
                PROF_ADD(6, t_p1, t_p);
            } else {
                // ---- S1b: stage, mark
                asm volatile("; MARK_S1B_BEGIN");
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    if ((uint32_t)i < nv) {
                        const uint32_t slot = (bfl[i] >> 8) & 0xffu;
                        if ((bfl[i] >> 16) & 1u) *reinterpret_cast<uint2 *>(&S.stage[slot * (uint32_t)R_SS + 2u * lane]) = make_uint2(r0[i], r1[i]);
                        else S.stage[slot * (uint32_t)R_SS + lane] = r0[i] | r1[i] << 16;
                    }
                }
                uint32_t x0[RB], x1[RB];  // id - tlo
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    x0[i] = bmin[i] + r0[i];
                    x1[i] = bmin[i] + r1[i];
                }
                if (!allfast) {  // raw (width 32), wide-delta or byte-packed tail blocks: generic, synchronous decode
#pragma nounroll
                    for (uint32_t i = 0; i < nv; ++i) {
                        const uint32_t e = (wave - 1u) + (RNW - 1) * i;
                        const uint32_t fl = uni(S.pa[buf][e].y);
                        if (!((fl >> 17) & 1u)) {
                            const uint4 cc = uni4(S.pm[buf][e]);
                            const uint32_t md = (cc.w >> 8) & 0xff, n = cc.w & 0xff, slot = (fl >> 8) & 0xffu;
                            VCHK(8ull * cc.z < ix.blob_bytes && ((fl >> 18) & 1u) && e < np, 33, cc.z);
#ifdef VBM25_CHECK
                            if (8ull * cc.z >= ix.blob_bytes) continue;
#endif
                            uint32_t a0, a1;
                            decode_doc_ids(ix.blob + 8ull * cc.z, md, n, cc.x, lane, a0, a1);
                            const bool in0 = 2 * lane < n, in1 = 2 * lane + 1 < n;
                            // entries past the block's end: staged above every id of the block (S2 checks the index), marked
                            // far outside the tile, one distinct id each
                            if ((fl >> 16) & 1u)
                                *reinterpret_cast<uint2 *>(&S.stage[slot * (uint32_t)R_SS + 2u * lane]) = make_uint2(in0 ? a0 - cc.x : NONE32, in1 ? a1 - cc.x : NONE32);
                            else
                                S.stage[slot * (uint32_t)R_SS + lane] = (in0 ? a0 - cc.x : 0xffffu) | (in1 ? a1 - cc.x : 0xffffu) << 16;
                            const uint32_t y0 = in0 ? a0 - tlo : 0xffffff00u + 2 * lane, y1 = in1 ? a1 - tlo : 0xffffff01u + 2 * lane;
#pragma unroll
                            for (int j = 0; j < RB; ++j) {
                                x0[j] = i == (uint32_t)j ? y0 : x0[j];
                                x1[j] = i == (uint32_t)j ? y1 : x1[j];
                            }
                        }
                    }
                }
                // mark.  Postings of a straddling block that belong to a neighbour tile mark a bit too (no range test
                // here); the event / cold paths test the range.  All the atomics of the wave's blocks are in flight together.
                uint32_t ev = 0;  // bit 2 i + j: posting j of entry i is a second arrival
                asm volatile("; MARK_MARKS_BEGIN");
#pragma unroll
                for (int h = 0; h < RB; h += 4) {  // (four blocks' atomics in flight together)
                    uint32_t m0[4], m1[4], o0[4], o1[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int i = h + g;
                        if (i < RB) {
                            const uint32_t t0 = x0[i] >> 17, t1 = x1[i] >> 17;
                            m0[g] = (1u << (x0[i] & 31u)) | (1u << ((x0[i] + t0 * 5u + 1u) & 31u)) | (1u << ((x0[i] + 13u) & 31u));
                            m1[g] = (1u << (x1[i] & 31u)) | (1u << ((x1[i] + t1 * 5u + 1u) & 31u)) | (1u << ((x1[i] + 13u) & 31u));
                            if ((uint32_t)i >= nv) m0[g] = m1[g] = 0;  // (an unused entry marks nothing; uniform)
                            o0[g] = atomicOr(&S.bm[(x0[i] >> 5) & (R_BM_WORDS - 1)], m0[g]);
                            o1[g] = atomicOr(&S.bm[(x1[i] >> 5) & (R_BM_WORDS - 1)], m1[g]);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int i = h + g;
                        if (i < RB)
                            ev |= ((o0[g] & m0[g]) == m0[g] && m0[g] != 0u ? 1u : 0u) << (2 * i) | ((o1[g] & m1[g]) == m1[g] && m1[g] != 0u ? 1u : 0u) << (2 * i + 1);
                    }
                }
                asm volatile("; MARK_MARKS_END");
                // second arrivals -> this wave's event list -> insert passes into the hash set (one row per document)
                if (__ballot(ev != 0)) {
                    uint32_t cnt = 0;
                    auto insert_events = [&]() {
                        __builtin_amdgcn_wave_barrier();
                        for (uint32_t j = lane; j < cnt; j += 64) {
                            const uint32_t d = tlo + S.hits[wave][j];
                            uint32_t slot = (d * 0x9E3779B1u) >> (32 - R_HS_LOG2);
                            for (uint32_t probes = 0;; ++probes) {
                                const uint32_t prev = atomicCAS(&S.hkeys[slot], EMPTY, d);
                                if (prev == EMPTY) {
                                    const uint32_t r = atomicAdd(&S.nrows[buf], 1u);
                                    if (r < ROWS + (uint32_t)R_XROWS) S.mdoc[par][r] = d;
                                    else S.fail = 1;
                                    if (r >= ROWS) late_push(d);  // found like a row (done bits), scored by lookups
                                    break;
                                }
                                if (prev == d) break;
                                slot = (slot + 1) & (R_HS - 1);
                                if (probes >= 128u) {
                                    S.fail = 1;
                                    break;
                                }
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        cnt = 0;
                    };
                    uint32_t mask = ev;
                    while (__ballot(mask != 0)) {
                        const bool has = mask != 0;
                        const uint32_t b = has ? (uint32_t)__ffs((int)mask) - 1u : 0u;
                        mask &= mask - 1u;
                        // the id comes back from the wave's own stage row (one LDS read instead of a 16-way select)
                        const uint32_t e = (wave - 1u) + (RNW - 1) * (b >> 1);
                        const uint32_t fl = S.pa[buf][e].y;
                        const uint32_t slot = (fl >> 8) & 0xffu, idx = 2u * lane + (b & 1u);
                        const uint32_t rel = (fl >> 16) & 1u ? S.stage[slot * (uint32_t)R_SS + idx]
                                                              : (uint32_t)reinterpret_cast<const uint16_t *>(&S.stage[slot * (uint32_t)R_SS])[idx];
                        const uint32_t x = S.pm[buf][e].x - tlo + rel;
                        const bool ok = has && x < span;  // (postings of the neighbour tiles, entries past a tail block's end)
                        const unsigned long long om = __ballot(ok);
                        if (ok) S.hits[wave][cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u))] = x;
                        cnt += (uint32_t)__popcll(om);
                        if (cnt > (uint32_t)R_EVENTS - 64u) insert_events();  // (the next pass adds at most 64)
                    }
                    if (cnt) insert_events();
                }
                asm volatile("; MARK_EVENTS_END");
                PROF_T(t_s);
                PROF_ADD(1, t_a0, t_a1);
                PROF_ADD(6, t_a1, t_s);
                hits_consume();
                // the wave's raw slots are free: request the next tile's id words as soon as its plan is there
                if (uni(S.plan_seq) == tile + 1u) {
                    dma_issue((tile + 1) % R_PLAN_RING);
                    dma_done = true;
                }
#ifdef VBM25_PROFILE
                prof[0] += 1;
                prof[13] += nh;
#endif
            }
            if (lane == 0) S.hcnt[wave] = 0;
            PROF_T(t_b);
            lds_barrier();  // ---- A: every mark, staged id and row of the tile is in LDS; contributions of the previous tile; the next plan
            PROF_T(t_c);
            PROF_ADD(2, t_b, t_c);
            if (uni(S.fail)) {
                failed = true;
                break;
            }
            if (wave != 0 && !dma_done) dma_issue((tile + 1) % R_PLAN_RING);

            // ---- S2: wipe the filter and the hash set; rows x terms: find the postings
            const uint32_t nm = min(uni(S.nrows[buf]), ROWS + (uint32_t)R_XROWS);
#pragma unroll
            for (int i = 0; i < R_BM_WORDS / 4 / RWG; ++i)
                reinterpret_cast<uint4 *>(S.bm)[tid + i * RWG] = make_uint4(0, 0, 0, 0);
            if (tid < R_HS / 4) reinterpret_cast<uint4 *>(S.hkeys)[tid] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
            if (nm) {
                // tasks = rows x the query's terms, packed: waves beyond nm * mq tasks skip the phase
                const uint32_t inv_mq = (65536u + mq - 1u) / mq;  // p / mq == (p * inv_mq) >> 16 for p < 4096
                uint32_t hpos = 0;  // this wave's hit records so far (uniform)
                for (uint32_t p0 = wave * 64u; p0 < nm * mq; p0 += RWG) {
                    const uint32_t p = p0 + lane;
                    bool found = false;
                    uint32_t rec = 0;
                    if (p < nm * mq) {
                        const uint32_t r = (p * inv_mq) >> 16, t = p - r * mq;
                        const uint32_t d = S.mdoc[par][r];
                        uint32_t eb = S.ptb[buf][t], len = S.ptb[buf][t + 1] - eb;
                        if (len != 0) {
                            // last entry of the term with min_doc <= d: 4-ary (three independent reads per round trip)
                            while (len > 1) {
                                const uint32_t qn = (len + 3u) >> 2;
                                const uint32_t p1 = qn, p2 = 2u * qn, p3 = 3u * qn;
                                const uint32_t m1 = p1 < len ? S.pm[buf][eb + p1].x : NONE32;
                                const uint32_t m2 = p2 < len ? S.pm[buf][eb + p2].x : NONE32;
                                const uint32_t m3 = p3 < len ? S.pm[buf][eb + p3].x : NONE32;
                                const uint32_t c = (p1 < len && m1 <= d ? 1u : 0u) + (p2 < len && m2 <= d ? 1u : 0u) + (p3 < len && m3 <= d ? 1u : 0u);
                                eb += c * qn;
                                len = c == 3u ? len - p3 : min(qn, len - c * qn);
                            }
                            const uint4 sj = S.pm[buf][eb];
                            if (d >= sj.x && d <= sj.y) {
                                const uint32_t fl = S.pa[buf][eb].y, slot = (fl >> 8) & 0xffu, rel = d - sj.x;
                                // first staged id >= rel among the block's 128 (sorted, padded above): 4-ary as well
                                uint32_t idx = 0;
                                if ((fl >> 16) & 1u) {
                                    const uint32_t *sb = &S.stage[slot * (uint32_t)R_SS];
                                    idx = 32u * ((sb[31] < rel ? 1u : 0u) + (sb[63] < rel ? 1u : 0u) + (sb[95] < rel ? 1u : 0u));
                                    idx += 8u * ((sb[idx + 7] < rel ? 1u : 0u) + (sb[idx + 15] < rel ? 1u : 0u) + (sb[idx + 23] < rel ? 1u : 0u));
                                    idx += 2u * ((sb[idx + 1] < rel ? 1u : 0u) + (sb[idx + 3] < rel ? 1u : 0u) + (sb[idx + 5] < rel ? 1u : 0u));
                                    idx += sb[idx] < rel ? 1u : 0u;
                                    found = sb[idx] == rel;
                                } else {
                                    const uint16_t *sb = reinterpret_cast<const uint16_t *>(&S.stage[slot * (uint32_t)R_SS]);
                                    idx = 32u * (((uint32_t)sb[31] < rel ? 1u : 0u) + ((uint32_t)sb[63] < rel ? 1u : 0u) + ((uint32_t)sb[95] < rel ? 1u : 0u));
                                    idx += 8u * (((uint32_t)sb[idx + 7] < rel ? 1u : 0u) + ((uint32_t)sb[idx + 15] < rel ? 1u : 0u) + ((uint32_t)sb[idx + 23] < rel ? 1u : 0u));
                                    idx += 2u * (((uint32_t)sb[idx + 1] < rel ? 1u : 0u) + ((uint32_t)sb[idx + 3] < rel ? 1u : 0u) + ((uint32_t)sb[idx + 5] < rel ? 1u : 0u));
                                    idx += (uint32_t)sb[idx] < rel ? 1u : 0u;
                                    found = (uint32_t)sb[idx] == rel;
                                }
                                found = found && idx < (sj.w & 0xffu);  // (not an entry past a tail block's end)
                                if (found) {
                                    atomicOr(&S.done[eb * 4 + (idx >> 5)], 1u << (idx & 31));
                                    rec = r << 16 | eb << 8 | idx;
                                    found = r < ROWS;  // a document beyond the rows: done bit only, the late list scores it
                                }
                            }
                        }
                    }
                    const unsigned long long fm = __ballot(found);
                    if (fm) {
                        const uint32_t pos = hpos + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                        if (found) {
                            if (pos < (uint32_t)R_HITS) S.hits[wave][pos] = rec;
                            else S.fail = 2;  // (cannot happen: at most 1024 tasks with a row, two rounds of 64 per wave)
                        }
                        hpos += (uint32_t)__popcll(fm);
                    }
                }
                if (lane == 0) S.hcnt[wave] = min(hpos, (uint32_t)R_HITS);
            }
            PROF_T(t_d);
            PROF_ADD(3, t_c, t_d);

            // ---- S3 (all waves, ROWS / 8 rows each): rows of the previous tile -> documents -> pool.  Their contributions were
            // written before barrier A.
            {
                const uint32_t nmp = min(uni(S.nrows[pbuf]), ROWS);
                const uint32_t ppne = uni(S.hdr[pbuf].w) & 0xffu;
                if (tid == 0) {
                    // what the previous tile's... this tile's tail decides by, uniformly: an upper bound of the pool after the
                    // rows' pushes, the late list, whether a threshold exists
                    S.pool_snap = S.pool_n + (tile != 0 ? nmp : 0u);
                    S.late_snap = S.nlate;
                    S.theta_zero = S.theta == 0ull ? 1u : 0u;
                    S.rows_seen = nm;
                    S.nrows[(tile + 1) % 3u] = 0;  // (last read by the S3 of the previous iteration)
                }
                constexpr uint32_t RPW = ROWS / RNW;
                if (tile != 0 && nmp > wave * RPW) {
                    const double pnes = ppne ? S.t_cum[ppne] : 0.0;
                    uint32_t emask = 0xffffffffu;  // bit t: term t is essential
                    if (ppne) {
                        emask = 0;
                        for (uint32_t t = 0; t < mq; ++t) emask |= ((uint32_t)S.t_rank[t] >= ppne ? 1u : 0u) << t;
                    }
                    const uint32_t r = wave * RPW + lane;
                    bool has = lane < RPW && r < nmp;
                    double acc = 0.0;
                    uint32_t d = 0;
                    if (has) {
                        d = S.mdoc[par ^ 1u][r];
                        for (uint32_t t = 0; t < mq; ++t) acc += S.contrib[(r << LRT) + t];  // ascending key order; absent terms add 0.0
                    }
                    const bool mine = has;
                    VCHK(!has || acc * S.hscale < (double)CUR_HB, 35, r);
                    if (ppne != 0) acc = complete(has, d, has ? r : NONE32, 0, 0.0, acc, emask, pnes);  // (reads the row's contributions)
                    pool_push(has, acc, d);
                    if (mine)
                        for (uint32_t t = 0; t < mq; ++t) S.contrib[(r << LRT) + t] = 0.0;
#ifdef VBM25_PROFILE
                    if (wave == 0) prof[10] += nmp;
#endif
                }
            }
            PROF_T(t_e);
            PROF_ADD(3, t_c, t_e);
            np_prev = np;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (an LDS-DMA in flight must not land in the next item's LDS)
        __syncthreads();
        if (!failed && uni(S.nlate)) {
            shrink_pool();  // room for the late documents' scores
            if (!uni(S.fail)) flush_late();
        }
        if (uni(S.fail)) failed = true;

#ifdef VBM25_PROFILE
        prof[12] += 1;
        prof[9] += __builtin_readcyclecounter() - t_loop;
#endif
        // ---- item result: the pool cut into one sorted list per wave (merge_kernel / the fused merge take them from there)
        RegTopK<RK> rtop;
        rtop.init();
        if (!failed) {
            const uint32_t pn = min(uni(S.pool_n), POOL);
            const unsigned long long thb = theta_now();
            for (uint32_t base = wave * 64u; base < pn; base += RWG) {
                const uint32_t i = base + lane;
                bool has = i < pn;
                double sc = 0.0;
                uint32_t d = 0;
                if (has) {
                    const unsigned long long sb = S.pool_s[i];
                    sc = __longlong_as_double((long long)sb);
                    d = S.pool_d[i];
                    VCHK(sc * S.hscale < (double)CUR_HB, 34, i | pn << 16);
                    has = sb != 0ull && sb >= thb;
                }
                has = has && (rtop.cnt < k || better(sc, d, rtop.kth_s, rtop.kth_d));
                if (__ballot(has)) rtop.offer(has, sc, d, k, lane);
            }
            VCHK(rtop.cnt < k || rtop.kth_s * S.hscale < (double)CUR_HB, 28, rtop.kth_d);
            if (rtop.cnt >= k && lane == 0) atomicMax(&bt.theta[q], (unsigned long long)__double_as_longlong(rtop.kth_s));
        }
        const uint32_t n = failed ? 0u : rtop.cnt;
        const size_t list = (size_t)item * bt.lpi + wave;
#pragma unroll
        for (int r = 0; r < RK; ++r)
            if (r * 64 + lane < n) {
                bt.res_score[list * k + r * 64 + lane] = rtop.score[r];
                bt.res_doc[list * k + r * 64 + lane] = rtop.doc[r];
            }
        if (lane == 0) {
            bt.res_cnt[list] = n;
            if (wave == 0) bt.item_failed[item] = failed ? (S.fail | 0x100u) : 0u;
        }
        if constexpr (FUSED) {
            // ---- the last workgroup of the query merges its lists (merge.h's job) and leaves the per-launch state clean.
            // One item per query: the eight lists are this workgroup's own and travel through LDS.  Several items: an
            // atomic counter per query finds the last workgroup; the other workgroups' lists are read with loads that
            // bypass this CU's vector cache, all entries of 16 lists at a time (one round trip, not one per list).
            uint32_t *tmp = reinterpret_cast<uint32_t *>(S.raw);  // raw + stage = 28 KB, free between items (8 x 3 x KMAX words; 16 lists x k words)
            static_assert(offsetof(Lds, stage) == offsetof(Lds, raw) + sizeof(S.raw), "raw and stage are one region here");
            static_assert(sizeof(S.raw) + sizeof(S.stage) >= 4u * 8u * 3u * (uint32_t)KMAX && sizeof(S.raw) + sizeof(S.stage) >= 4u * 16u * (uint32_t)KMAX,
                          "fused merge scratch");
            bool last = true;
            if (fused_g == 1u) {
                __syncthreads();
                const uint32_t base = wave * 3u * (uint32_t)KMAX;
#pragma unroll
                for (int r = 0; r < RK; ++r)
                    if (r * 64 + lane < n) {
                        tmp[base + r * 64 + lane] = (uint32_t)__double2loint(rtop.score[r]);
                        tmp[base + KMAX + r * 64 + lane] = (uint32_t)__double2hiint(rtop.score[r]);
                        tmp[base + 2 * KMAX + r * 64 + lane] = rtop.doc[r];
                    }
                if (lane == 0) S.hcnt[wave] = n;
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
                        const uint32_t cnt = uni(S.hcnt[w]), base = w * 3u * (uint32_t)KMAX;
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
                        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);  // <= 16 k words
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
                const uint32_t nhit = rtop.cnt;
                const bool failed_any = __ballot(any_failed != 0) != 0ull;
#pragma unroll
                for (int r = 0; r < RK; ++r)
                    if (r * 64 + lane < nhit) {
                        const uint32_t d = rtop.doc[r];
                        const uint16_t *pl = ix.doc_payload + 3ull * d;
                        unsigned long long *out = reinterpret_cast<unsigned long long *>(bt.hits + (size_t)q * k + r * 64 + lane);
                        out[0] = (unsigned long long)__double_as_longlong(rtop.score[r]);
                        out[1] = (unsigned long long)d | (unsigned long long)pl[0] << 32 | (unsigned long long)pl[1] << 48;
                        out[2] = (unsigned long long)pl[2];
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) hrow[4 * lane + i] = 0;
                if (lane == 0) {
                    bt.n_hits[q] = failed_any ? NONE32 : nhit;  // NONE32: an item needs scan_many_kernel -- the host re-runs the batch on the general route
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
