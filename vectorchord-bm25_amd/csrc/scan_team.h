// scan_team.h -- scan_team_kernel: sparse queries of <= 16 indexed terms, k <= REG_K (the dominant kernel: C3).
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25.
//
// Replaces the WAND main loop of search.rs:149-280 for the sealed segment.  The formulation was chosen with
// tools/ubench/mark_ceiling.hip (profiles/r4_mark_ceiling.txt): planning + plane loads + marks alone run at 0.28 of the
// HBM roofline in this form, 0.20-0.23 with wave-private bitmaps (LDS caps those at two waves per SIMD and the
// instruction stream, not memory, is the bound) -- the round-3 kernel did everything at 0.145.
//
// One workgroup = a TEAM of waves works on one item (query x document range), window by window.  A window is
// TEAM x 65536 documents (x 2^shift for items so sparse that a window would hold a handful of blocks) and has ONE exact
// bitmap in LDS, one bit per document (per 2^shift documents).
//
//   ownership   block j of a term belongs to wave j mod TEAM: every wave plans only its own residue class with its 64
//               lanes (lane = term x 4 candidate blocks; blocks of a term ascend, so the blocks inside the window are a
//               prefix of the candidates) -- no planner wave, no shared plan, no barrier to publish one.
//   marks       a posting's two ids come from ONE coalesced word of post_rel16 (device_types.h); each id is marked with
//               one returning LDS atomic.  A mark that was already there = a SECOND ARRIVAL: the document may be in two
//               lists -- seen by whichever wave comes second, in any order.  Blocks that straddle a window boundary are
//               read by both windows (7 % of the reads at TEAM = 8, 15 % at 4) and their postings outside the window
//               mark nothing.  Full groups of TM_G blocks take the branch-free path (the next group's loads in flight);
//               tails, wide or raw blocks, blocks whose upper bound reaches the threshold and the remainder of a round
//               take the general path, one block at a time.
//   candidates  second arrivals and -- from the blocks whose upper bound (search.rs:377-380, once per index) reaches the
//               threshold (search.rs:203) -- every posting that could reach it alone go to the wave's candidate list.
//   completion  a candidate document is scored from scratch, whoever found it: lanes = (candidate, term).  The term's
//               block that holds the document comes from the window's DIRECTORY in LDS (every wave enters the first
//               documents of the blocks it plans: [window mod 3][term][block mod 16]; Cursor::seek_block, search.rs:412-431,
//               without a single global load), the posting from a two-level search of the block's 16-bit ids (blk_piv:
//               every 16th id of the block, then 32 bytes of the plane), its tf / fieldnorm from post_tfn, then
//               Cache::evaluate (bm25.rs:355-358) and the sum in ascending key order (evaluate.rs:43-72) -> the wave's
//               register top-k.  These are four dependent round trips to memory; done on the spot they doubled the kernel's
//               time (every wave of a team waits for the slowest).  So completion is a PIPELINE: a batch advances ONE stage
//               after every group of marks -- its loads were issued a group earlier and return in order before the
//               group's own plane words, which the marks wait for anyway.  Candidates of window n are completed during
//               windows n + 1 and n + 2 (the directory keeps three windows).  A document found twice (three lists; a cold
//               posting that is also a second arrival; two waves) is scored twice to the same bits: RegTopK and
//               merge_kernel drop the repetition.  Lookups the directory cannot serve (more than 16 blocks of a term in a
//               window, a wave whose ring is full of the current window's candidates) search blk_max_doc in global memory.
//   life cycle  of the bitmap, the only thing the team synchronises for: every mark of window n before it is wiped, the
//               wipe before the marks of n + 1.  Two LDS counters, arrive early / wait late, no s_barrier inside an item.
//
// There is no overflow mode: a full candidate ring is drained on the spot, a round takes what its lanes hold and the
// next round takes the rest; nothing is handed to scan_many_kernel.
//
// Threshold: theta0 = the largest, over the query's terms, of the k-th largest block maximum of the term
// (term_kth_ub, derived at index creation; every block maximum is the score of a posting of its block, documents of one
// term are distinct, and a document's score is at least any of its postings') -- a lower bound of the final k-th
// score before the first posting is read; then the waves' k-th scores through LDS and the query's 64-bit atomicMax
// word.  Filtering on score < threshold is exact: ties are kept.

constexpr int TM_TQ = 4;         // candidate blocks per term and round (16 terms x 4 = the 64 lanes)
constexpr int TM_G = 4;          // blocks per group of the branch-free path
constexpr int TM_LIST = 64;      // ring of candidate documents per wave (a power of two)
constexpr int TM_KTH = 9;        // term_kth_ub entries per term: the 2^i-th largest block maximum, i = 0..8
constexpr int TM_DIR = 16;       // directory slots per term and window

template <int TEAM>
struct TeamLds {
    uint32_t bm[TEAM * 2048];
    uint32_t dir_min[3][16][TM_DIR];  // [window mod 3][term][block mod 16]: first document of the block
    uint32_t dir_flag[3][16];         // per term, two bits per slot: 1 = the block has a post_rel16 word, 2 = a post_tfn word
    uint32_t dir_lo[3][16], dir_hi[3][16];  // smallest / largest block of the term inside the window (lo > hi: none)
    uint32_t cand[TEAM][TM_LIST];     // per wave: ring of candidate documents
    uint32_t scr[TEAM][128];          // per wave: ids of a block decoded for a lookup / the contributions of a batch
    unsigned long long theta;         // bits of a lower bound of the query's k-th best score
    uint32_t sync[2];
    uint32_t item, fail;
};

// First block of [b0, b1) whose max_doc >= d (b1 if none): a float guess by interpolation, gallop, bisection.
__device__ __forceinline__ uint32_t tm_first_block_ge(const uint32_t *blk_max_doc, uint32_t b0, uint32_t b1, uint32_t d, float inv_docs) {
    uint32_t lo_b = b0, hi_b = b1;
    if (b1 > b0) {
        uint32_t g = b0 + (uint32_t)((float)(b1 - b0) * ((float)d * inv_docs));
        if (g >= b1) g = b1 - 1;
        if (blk_max_doc[g] < d) {
            lo_b = g + 1;
            for (uint32_t step = 1; lo_b < hi_b; step *= 4) {
                const uint32_t p = min(lo_b + step - 1, hi_b - 1);
                if (blk_max_doc[p] < d) lo_b = p + 1;
                else {
                    hi_b = p;
                    break;
                }
            }
        } else {
            hi_b = g;
            for (uint32_t step = 1; lo_b < hi_b; step *= 4) {
                const uint32_t p = hi_b - lo_b >= step ? hi_b - step : lo_b;
                if (blk_max_doc[p] >= d) hi_b = p;
                else {
                    lo_b = p + 1;
                    break;
                }
            }
        }
        while (lo_b < hi_b) {
            const uint32_t mid = (lo_b + hi_b) >> 1;
            if (blk_max_doc[mid] < d) lo_b = mid + 1; else hi_b = mid;
        }
    }
    return lo_b;
}

// (macros, not lambdas: with this many by-reference closures in one function the optimiser leaves every captured local --
// and a copy of both argument structs -- in scratch memory)
#define TM_THETA_NOW()                                                                                          \
    ({                                                                                                          \
        unsigned long long th_ = S.theta;                                                                       \
        th_ = ((unsigned long long)uni((uint32_t)(th_ >> 32)) << 32) | uni((uint32_t)th_);                      \
        th_ > theta0 ? th_ : theta0;                                                                            \
    })
// the plane words of the next TM_G blocks of m_fast; the last group is padded: every id out of the window (lane 0's block is
// read again)
#define TM_ISSUE()                                                                                              \
    do {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < TM_G; ++i_) {                                                   \
            const bool has_ = m_fast != 0ull;                                                                   \
            const int sl_ = has_ ? __ffsll((long long)m_fast) - 1 : 0;                                          \
            m_fast &= m_fast - 1;                                                                               \
            const uint32_t blk_ = (uint32_t)__builtin_amdgcn_readlane((int)v_blk, sl_);                         \
            const uint32_t dd_ = (uint32_t)__builtin_amdgcn_readlane((int)v_delta, sl_);                        \
            dl[i_] = has_ ? dd_ : 0x80000000u;                                                                  \
            w[i_] = ix.post_rel16[64ull * blk_ + lane];                                                         \
        }                                                                                                       \
    } while (false)
#define TM_AFTER_GROUPS() (m_slow ? (uint32_t)PH_SLOW : again ? (uint32_t)PH_ROUND : (uint32_t)PH_WIN_A)

// The kernel's arguments as they lie in the kernarg segment.  The pointers that only the item setup, the item's end and the
// rare paths need are read from there where they are used (cold_args): kept in SGPRs for the whole kernel they push the loop's
// uniform state into VGPRs and the VGPRs into scratch.
struct TeamArgs {
    DevIndex ix;
    DevBatch bt;
};
typedef const __attribute__((address_space(4))) TeamArgs *TeamArgsP;
__device__ __forceinline__ TeamArgsP cold_args() {
    TeamArgsP p = (TeamArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));  // opaque: the loads stay where they are written
    return p;
}

template <int KMAX, int TEAM>
__global__ void __launch_bounds__(TEAM * 64, 4) scan_team_kernel(DevIndex ix, DevBatch bt) {
    static_assert(KMAX <= REG_K, "register top-k only");
    static_assert(TEAM == 4 || TEAM == 8, "two or four workgroups per CU");
    constexpr int RK = KMAX / 64;
    constexpr uint32_t BITS = (uint32_t)TEAM << 16, WORDS = BITS / 32;
    __shared__ TeamLds<TEAM> S;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t k = bt.k;
    const uint32_t n_items = *cold_args()->bt.n_items;
    uint32_t *const cand = S.cand[wave];
    uint32_t *const scr = S.scr[wave];
    const uint32_t slot_t = lane / TM_TQ, off = lane % TM_TQ;
    const float inv_docs = 1.0f / (float)ix.n_docs;
    uint32_t kidx = 0;  // term_kth_ub entry: the smallest 2^i >= k
    while ((1u << kidx) < k) ++kidx;

    for (uint32_t i = tid; i < WORDS; i += TEAM * 64) S.bm[i] = 0;
    if (tid < 2) S.sync[tid] = 0;
    uint32_t epoch = 0;  // windows this workgroup has finished: the counters take TEAM arrivals per window
    auto arrive = [&](int which) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) atomicAdd(&S.sync[which], 1u);
    };
    auto wait = [&](int which) {
        const uint32_t target = (epoch + 1u) * TEAM;
        while (__hip_atomic_load(&S.sync[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };

    for (;;) {
        __syncthreads();  // previous item: every wave is past its last wait and has drained its candidates
        if (tid == 0) {
            const TeamArgsP ca = cold_args();
            const uint32_t drawn = atomicAdd(&ca->bt.work_ctr[0], 1u);
            S.item = drawn < n_items ? ca->bt.item_order[drawn] : NONE32;  // plan_kernel's order: longest first
            S.theta = 0;
            S.fail = 0;
        }
        if (tid < 48) {  // the three directories: no block yet
            (&S.dir_lo[0][0])[tid] = NONE32;
            (&S.dir_hi[0][0])[tid] = 0;
            (&S.dir_flag[0][0])[tid] = 0;
        }
        __syncthreads();
        const uint32_t item = uni(S.item);
        if (item == NONE32) break;
        const TeamArgsP ca = cold_args();
        const Item it = ca->bt.items[item];
        if (it.m > 16u || it.m == 0u) continue;  // more terms or dense (ITEM_DENSE): the other kernels'
        const uint32_t q = uni(it.q), lo = uni(it.doc_lo), hi = uni(it.doc_hi);

        // ---- the query's indexed terms: lane t = term t (every wave for itself: a handful of loads, no hand-off)
        uint32_t m = 0, term = NONE32;
        {
            const uint32_t qb = uni(ca->bt.q_off[q]), qe = uni(ca->bt.q_off[q + 1]);
            if (qe - qb <= 64) {  // one load per lane, compaction of the indexed terms through LDS
                const uint32_t tt = lane < qe - qb ? ca->bt.term_ids[qb + lane] : NONE32;
                const bool ok = tt < ix.n_terms;  // search.rs:59-61
                const unsigned long long okm = __ballot(ok);
                if (ok) scr[__builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = tt;
                __builtin_amdgcn_wave_barrier();
                m = (uint32_t)__popcll(okm);
                if (lane < m) term = scr[lane];
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
        uint32_t r_b0 = 0, r_b1 = 0, r_df = 0;
        double r_s0 = 0.0, kth = 0.0;
        if (act) {
            r_b0 = ca->ix.term_first_block[term];
            r_b1 = ca->ix.term_first_block[term + 1];
            r_s0 = ca->ix.term_s0[term];
            r_df = ca->ix.term_df[term];
            const double *kub = ca->ix.term_kth_ub;
            if (kub) kth = kub[(size_t)term * TM_KTH + kidx];
        }
        // theta0 and the item's postings (scores are >= 0: fmax over the lanes)
        unsigned long long postings = r_df;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            kth = fmax(kth, __shfl_xor(kth, o));
            postings += __shfl_xor(postings, o);
        }
        const unsigned long long theta0 = (unsigned long long)__double_as_longlong(kth);
        if (wave == 0 && lane == 0 && theta0) atomicMax(&bt.theta[q], theta0);
        // documents per bitmap bit: a window should hold a few blocks per wave
        uint32_t shift = 0;
        {
            float b = (float)(postings >> 7) * ((float)BITS * inv_docs);
            while (b < (float)(2 * TEAM) && shift < (TEAM == 8 ? 12u : 13u)) {
                b *= 2.0f;
                ++shift;
            }
        }
        const uint32_t W = BITS << shift;  // documents per window (<= 2^31)

        // ---- this wave's cursor per term: its first block (j mod TEAM == wave) whose last document is >= lo
        const bool slot_act = slot_t < m;
        uint32_t cur = 0, end = 0;
        {
            const uint32_t b0s = (uint32_t)__shfl((int)r_b0, (int)slot_t), b1s = (uint32_t)__shfl((int)r_b1, (int)slot_t);
            if (slot_act) {
                const uint32_t f = lo ? tm_first_block_ge(ca->ix.blk_max_doc, b0s, b1s, lo, inv_docs) : b0s;
                cur = f + ((wave + (uint32_t)TEAM - f % (uint32_t)TEAM) % (uint32_t)TEAM);
                end = b1s;
            }
        }

        RegTopK<RK> rtop;
        rtop.init();
        unsigned long long published = 0;
        // ================================================================================================================
        // Completion pipeline.  The ring holds the candidates of up to three windows, oldest first: q0 of window w - 2, q1
        // of w - 1, q2 of the current window w (directories (w + 1) % 3, (w + 2) % 3, w % 3).  A batch = up to 64 / m
        // candidates of ONE window x the m terms; q2 is not touched while its window is open (the other waves are still
        // entering their blocks) unless the ring is full of it -- then the lookups go to global memory.
        // ================================================================================================================
        uint32_t q_head = 0, q0 = 0, q1 = 0, q2 = 0, wpar = 0;
        const uint32_t inv_m = (65536u + m - 1u) / m;  // p / m == (p * inv_m) >> 16 for p < 4096
        const uint32_t per_batch = 64u / m;
        uint32_t cs_stage = 0, cs_n = 0, cs_from = 0;  // uniform: stage, candidates of the batch, the run it was taken from (0 / 1 / 2)
        // per lane (= one (candidate, term) of the batch), slots reused from stage to stage:
        //   cs_d   the document            cs_j   the term's block that may hold it
        //   cs_x   the block's first document (stages 0-2), then the posting's index in the block
        //   cs_fl  1 plane word, 2 tfn word, 4 there is such a block, 8 look it up in global memory; bits 8.. = 16 x the id segment
        //   cs_v   the pivots (4 words), the 16 ids (8), the tf word, tf / fieldnorm, s1 (2)
        uint32_t cs_d = 0, cs_j = 0, cs_x = 0, cs_fl = 0;
        uint32_t cs_v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // ================================================================================================================
        // The wave's control flow is a state machine: every iteration does ONE unit of marking work (a round's plan, a group
        // of blocks, one general block, a step of the window's end) and then ONE stage of the completion pipeline -- so
        // the pipeline's code exists once, and "wait until the ring has room / the old window's candidates are done" is the
        // same iteration with no marking work in it.
        // ================================================================================================================
        enum : uint32_t { PH_OPEN, PH_ROUND, PH_GROUP, PH_SLOW, PH_STORE, PH_WIN_A, PH_WIN_B, PH_WIN_C, PH_END };
        uint32_t phase = lo < hi ? (uint32_t)PH_OPEN : (uint32_t)PH_END;
        uint32_t tlo = lo, thi = lo, span = 0;
        unsigned long long poll = 0;  // the query's threshold as the other workgroups see it: asked for when a window opens, used at its end
        unsigned long long m_fast = 0, m_slow = 0, m_cold = 0;
        uint32_t ngroups = 0, g = 0;
        bool again = false, failed = false;
        uint32_t v_delta = 0, v_blk = 0;
        uint32_t w[TM_G], dl[TM_G];
#pragma unroll
        for (int i = 0; i < TM_G; ++i) w[i] = dl[i] = 0;
        uint32_t st_d0 = 0, st_d1 = 0, st_stored = 0;  // a general block's candidates on their way into the ring
        unsigned long long st_e0 = 0, st_e1 = 0;
        // the first round's candidates
        uint32_t j = cur + (uint32_t)TEAM * off;
        bool valid = slot_act && j < end;
        uint4 meta = make_uint4(NONE32, 0, 0, 0);
        double ub = 0.0;
        if (valid) {
            meta = ix.blk_meta[j];
            ub = ix.blk_ub[j];
        }
        for (;;) {
            bool allow_q2 = false;
            const uint32_t room = (uint32_t)TM_LIST - (q0 + q1 + q2);
            if (phase == PH_OPEN) {
                thi = hi - tlo > W ? tlo + W : hi;
                span = thi - tlo;
                if (wave == 0) poll = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                phase = PH_ROUND;
            } else if (phase == PH_ROUND) {
                // ---- a round: up to TM_TQ blocks per term
                const double thd = __longlong_as_double((long long)TM_THETA_NOW());
                const bool in_win = valid && meta.x < thi;
                const bool has_plane = rel16_block(meta.x, meta.y, meta.w);
                const bool plane = shift == 0u && has_plane;
                const bool cold = ub >= thd;  // search.rs:203 per block (blk_ub carries its margin)
                m_fast = __ballot(in_win && plane && !cold);
                m_cold = __ballot(in_win && cold);
                m_slow = __ballot(in_win && !(plane && !cold));
                const unsigned long long done = __ballot(in_win && meta.y < thi);
                const uint32_t ndone = (uint32_t)__popcll((done >> (slot_t * TM_TQ)) & ((1ull << TM_TQ) - 1ull));
                again = __ballot(slot_act && ndone == (uint32_t)TM_TQ) != 0ull;
                if (in_win) {  // the window's directory: who holds which document range of the term
                    const uint32_t sl = j & (TM_DIR - 1);
                    S.dir_min[wpar][slot_t][sl] = meta.x;
                    const uint32_t fl = (has_plane ? 1u : 0u) | (tfn_block(meta.w) ? 2u : 0u);
                    if (fl) atomicOr(&S.dir_flag[wpar][slot_t], fl << (2u * sl));
                    atomicMin(&S.dir_lo[wpar][slot_t], j);
                    atomicMax(&S.dir_hi[wpar][slot_t], j);
                }
                v_delta = meta.x - tlo;
                v_blk = in_win ? j : 0u;  // (a padded group entry reads lane 0's block: a valid one)
                // the next round's candidates (their loads fly while this round is marked)
                cur += (uint32_t)TEAM * ndone;
                j = cur + (uint32_t)TEAM * off;
                valid = slot_act && j < end;
                meta = make_uint4(NONE32, 0, 0, 0);
                ub = 0.0;
                if (valid) {
                    meta = ix.blk_meta[j];
                    ub = ix.blk_ub[j];
                }
                ngroups = ((uint32_t)__popcll(m_fast) + TM_G - 1) / TM_G;
                g = 0;
                if (ngroups) {
                    TM_ISSUE();
                    phase = PH_GROUP;
                } else {
                    phase = TM_AFTER_GROUPS();
                }
            } else if (phase == PH_GROUP) {
                // ---- a group of TM_G blocks with a plane word: branch-free, the next group's words in flight
                if (room < 16u) {
                    allow_q2 = q0 == 0u && q1 == 0u;  // the ring first (its current-window candidates through global lookups)
                } else {
                    uint32_t x[2 * TM_G], mk[2 * TM_G], o[2 * TM_G];
#pragma unroll
                    for (int i = 0; i < TM_G; ++i) {
                        x[2 * i] = dl[i] + (w[i] & 0xffffu);
                        x[2 * i + 1] = dl[i] + (w[i] >> 16);
                    }
                    if (g + 1 < ngroups) TM_ISSUE();
#pragma unroll
                    for (int p = 0; p < 2 * TM_G; ++p) {
                        mk[p] = 1u << (x[p] & 31);
                        if (x[p] >= span) mk[p] = 0;  // another window's document: the atomic changes nothing
                        o[p] = atomicOr(&S.bm[(x[p] >> 5) & (WORDS - 1)], mk[p]);
                    }
                    unsigned long long dm[2 * TM_G], any = 0;
#pragma unroll
                    for (int p = 0; p < 2 * TM_G; ++p) {
                        dm[p] = __ballot((o[p] & mk[p]) != 0);
                        any |= dm[p];
                    }
                    if (any) {  // second arrivals -> the candidate ring
                        uint32_t c = 0;
#pragma unroll
                        for (int p = 0; p < 2 * TM_G; ++p) {
                            if (dm[p]) {
                                if ((o[p] & mk[p]) != 0) {
                                    const uint32_t at = c + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm[p] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm[p], 0u));
                                    if (at < room) cand[(q_head + q0 + q1 + q2 + at) & (TM_LIST - 1)] = tlo + x[p];
                                }
                                c += (uint32_t)__popcll(dm[p]);
                            }
                        }
                        if (c > room) failed = true;  // lists that intersect this densely: the item is scan_many_kernel's
                        q2 += min(c, room);
                    }
                    if (++g == ngroups) phase = TM_AFTER_GROUPS();
                }
            } else if (phase == PH_SLOW) {
                // ---- general path, one block: blocks without a plane word, windows of 2^shift documents per bit, blocks
                // whose upper bound reaches the threshold (every posting that could reach it alone is a candidate)
                const int sl = __ffsll((long long)m_slow) - 1;
                m_slow &= m_slow - 1;
                const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)v_blk, sl);
                const uint4 um = uni4(ix.blk_meta[blk]);
                const TeamArgsP cg = cold_args();
                const bool is_cold = (m_cold >> sl) & 1ull;
                const uint32_t un = um.w & 0xff, umd = (um.w >> 8) & 0xff, umt = (um.w >> 16) & 0xff;
                uint32_t d0, d1;
                if (rel16_block(um.x, um.y, um.w)) {
                    const uint32_t ww = ix.post_rel16[64ull * blk + lane];
                    d0 = um.x + (ww & 0xffffu);
                    d1 = um.x + (ww >> 16);
                } else {
                    decode_doc_ids(cg->ix.blob + 8ull * um.z, umd, un, um.x, lane, d0, d1);
                    if (2 * lane >= un) d0 = NONE32;
                    if (2 * lane + 1 >= un) d1 = NONE32;
                }
                const uint32_t x0 = d0 - tlo, x1 = d1 - tlo;
                const uint32_t g0 = x0 >> shift, g1 = x1 >> shift;
                const uint32_t k0 = x0 < span ? 1u << (g0 & 31) : 0u, k1 = x1 < span ? 1u << (g1 & 31) : 0u;
                const uint32_t o0 = atomicOr(&S.bm[(g0 >> 5) & (WORDS - 1)], k0);
                const uint32_t o1 = atomicOr(&S.bm[(g1 >> 5) & (WORDS - 1)], k1);
                bool f0 = (o0 & k0) != 0, f1 = (o1 & k1) != 0;
                if (is_cold) {
                    uint32_t t0, t1, n0, n1;
                    if (tfn_block(um.w)) {
                        const uint32_t ww = ix.post_tfn[64ull * blk + lane];
                        t0 = ww & 0xffu;
                        t1 = (ww >> 8) & 0xffu;
                        n0 = (ww >> 16) & 0xffu;
                        n1 = ww >> 24;
                    } else {
                        decode_fields(cg->ix.blob + 8ull * um.z + ((payload_bytes(umd, un) + 7u) & ~7u), umt, un, lane, t0, t1);
                        const uchar2 fn = reinterpret_cast<const uchar2 *>(cg->ix.post_fn + 128ull * blk)[lane];
                        n0 = fn.x;
                        n1 = fn.y;
                    }
                    // could the posting reach the threshold alone?  (tf s0) / (tf + s1) >= theta, without the division and
                    // with a margin: a candidate is scored exactly by its completion
                    const double s0t = readlane_f64(r_s0, (uint32_t)sl / TM_TQ);
                    const double thm = __longlong_as_double((long long)TM_THETA_NOW());
                    const double a0 = (double)t0, a1 = (double)t1;
                    f0 = f0 || (x0 < span && a0 * s0t * (1.0 + 1e-9) >= thm * (a0 + ix.s1[n0]));
                    f1 = f1 || (x1 < span && a1 * s0t * (1.0 + 1e-9) >= thm * (a1 + ix.s1[n1]));
                }
                st_e0 = __ballot(f0);
                st_e1 = __ballot(f1);
                if (st_e0 | st_e1) {
                    st_d0 = d0;
                    st_d1 = d1;
                    st_stored = 0;
                    phase = PH_STORE;
                } else {
                    phase = TM_AFTER_GROUPS();
                }
            } else if (phase == PH_STORE) {
                // ---- the block's candidates into the ring, as many as it has room for
                const uint32_t c0 = (uint32_t)__popcll(st_e0), c = c0 + (uint32_t)__popcll(st_e1);
                const bool f0 = (st_e0 >> lane) & 1ull, f1 = (st_e1 >> lane) & 1ull;
                if (f0) {
                    const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(st_e0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)st_e0, 0u));
                    if (at >= st_stored && at - st_stored < room) cand[(q_head + q0 + q1 + q2 + at - st_stored) & (TM_LIST - 1)] = st_d0;
                }
                if (f1) {
                    const uint32_t at = c0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(st_e1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)st_e1, 0u));
                    if (at >= st_stored && at - st_stored < room) cand[(q_head + q0 + q1 + q2 + at - st_stored) & (TM_LIST - 1)] = st_d1;
                }
                const uint32_t add = min(c - st_stored, room);
                q2 += add;
                st_stored += add;
                if (st_stored == c) phase = TM_AFTER_GROUPS();
                else allow_q2 = q0 == 0u && q1 == 0u;
            } else if (phase == PH_WIN_A) {
                // ---- the window's marks of this wave are in.  Before the arrival: the candidates of window w - 2 must be done
                // (their directory is the one that is reset below)
                if (!(q0 || (cs_stage && cs_from == 0u))) {
                    arrive(0);
                    if (wave == 0 && poll > TM_THETA_NOW() && lane == 0) atomicMax(&S.theta, poll);
                    phase = PH_WIN_B;
                }
            } else if (phase == PH_WIN_B) {
                wait(0);
                for (uint32_t i = tid; i < WORDS / 4; i += TEAM * 64) reinterpret_cast<uint4 *>(S.bm)[i] = make_uint4(0, 0, 0, 0);
                {
                    const uint32_t np = wpar == 2u ? 0u : wpar + 1u;  // the next window's directory
                    if (tid < 16) {
                        S.dir_lo[np][tid] = NONE32;
                        S.dir_hi[np][tid] = 0;
                        S.dir_flag[np][tid] = 0;
                    }
                }
                arrive(1);
                phase = PH_WIN_C;
            } else if (phase == PH_WIN_C) {
                wait(1);
                ++epoch;
                tlo = thi;
                wpar = wpar == 2u ? 0u : wpar + 1u;
                q0 = q1;  // (q0 was done before the arrival)
                q1 = q2;
                q2 = 0;
                if (cs_stage) cs_from = cs_from ? cs_from - 1u : 0u;
                phase = tlo < hi ? (uint32_t)PH_OPEN : (uint32_t)PH_END;
            } else {
                // ---- the item's last candidates: every directory is complete now
                if (!(cs_stage || q0 || q1 || q2)) break;
                allow_q2 = true;
            }
            // ---- one stage of the completion pipeline
            do {
            if (bt.team_dbg & 1u) {
                q_head += q0 + q1 + (allow_q2 ? q2 : 0u);
                q0 = q1 = 0;
                if (allow_q2) q2 = 0;
                break;
            }
            const uint32_t cs_ci = (lane * inv_m) >> 16, cs_t = lane - cs_ci * m;
            if (cs_stage == 0) {
                // ---- stage 0: a batch of the oldest window; the block of every (candidate, term) from the directory
                uint32_t n, par;
                bool glob = false;
                if (q0) {
                    n = min(q0, per_batch);
                    par = wpar + 1u;
                    cs_from = 0;
                } else if (q1) {
                    n = min(q1, per_batch);
                    par = wpar + 2u;
                    cs_from = 1;
                } else if (q2 && allow_q2) {
                    n = min(q2, per_batch);
                    par = wpar;
                    cs_from = 2;
                    glob = true;
                } else {
                    break;
                }
                if (par >= 3u) par -= 3u;
                cs_n = n;
                const bool task = cs_ci < n;
                cs_d = task ? cand[(q_head + cs_ci) & (TM_LIST - 1)] : 0u;
                cs_j = NONE32;
                cs_fl = 0;
                if (task && !glob) {
                    const uint32_t jl = S.dir_lo[par][cs_t], jh = S.dir_hi[par][cs_t];
                    if (jl <= jh) {
                        if (jh - jl >= (uint32_t)TM_DIR) {
                            cs_fl = 8;  // more blocks of the term in the window than the directory holds
                        } else {
                            const uint32_t *row = S.dir_min[par][cs_t];
                            const uint32_t cntj = jh - jl + 1u;
                            uint32_t pos = 0;  // the last block whose first document is <= d
#pragma unroll
                            for (uint32_t step = TM_DIR / 2; step > 0; step >>= 1)
                                if (pos + step < cntj && row[(jl + pos + step) & (TM_DIR - 1)] <= cs_d) pos += step;
                            const uint32_t mn = row[(jl + pos) & (TM_DIR - 1)];
                            if (mn <= cs_d) {
                                cs_j = jl + pos;
                                cs_x = mn;
                                cs_fl = 4u | ((S.dir_flag[par][cs_t] >> (2u * ((jl + pos) & (TM_DIR - 1)))) & 3u);
                            }
                        }
                    }
                } else if (task) {
                    cs_fl = 8;
                }
                if (__ballot((cs_fl & 8u) != 0)) {  // Cursor::seek_block in global memory (rare)
                    const uint32_t tb0 = (uint32_t)__shfl((int)r_b0, (int)cs_t), tb1 = (uint32_t)__shfl((int)r_b1, (int)cs_t);
                    if (cs_fl & 8u) {
                        cs_fl = 0;
                        const uint32_t jj = tm_first_block_ge(cold_args()->ix.blk_max_doc, tb0, tb1, cs_d, inv_docs);
                        if (jj < tb1) {
                            const uint4 mm = ix.blk_meta[jj];
                            if (mm.x <= cs_d) {
                                cs_j = jj;
                                cs_x = mm.x;
                                cs_fl = 4u | (rel16_block(mm.x, mm.y, mm.w) ? 1u : 0u) | (tfn_block(mm.w) ? 2u : 0u);
                            }
                        }
                    }
                }
                if ((cs_fl & 5u) == 5u) {
                    const uint4 pv = ix.blk_piv[cs_j];
                    cs_v[0] = pv.x;
                    cs_v[1] = pv.y;
                    cs_v[2] = pv.z;
                    cs_v[3] = pv.w;
                }
                cs_stage = 1;
            } else if (cs_stage == 1) {
                // ---- stage 1: the 16 ids that can hold the document
                if ((cs_fl & 5u) == 5u) {
                    const uint32_t r = cs_d - cs_x;
                    uint32_t seg = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) seg += ((cs_v[u] & 0xffffu) < r ? 1u : 0u) + ((cs_v[u] >> 16) < r ? 1u : 0u);
                    if (r > 0xffffu || seg > 7u) {
                        cs_fl = 0;  // the document lies behind the block's last id
                    } else {
                        cs_fl |= (16u * seg) << 8;
                        const uint4 *pp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(ix.post_rel16) + 128ull * cs_j + 16u * seg);
                        const uint4 a = pp[0], b = pp[1];
                        cs_v[0] = a.x;
                        cs_v[1] = a.y;
                        cs_v[2] = a.z;
                        cs_v[3] = a.w;
                        cs_v[4] = b.x;
                        cs_v[5] = b.y;
                        cs_v[6] = b.z;
                        cs_v[7] = b.w;
                    }
                }
                // blocks without a plane word (tails, raw, wide): the wave decodes each of them once
                bool pend = (cs_fl & 5u) == 4u;
                if (__ballot(pend)) {
                    const uint8_t *blob = cold_args()->ix.blob;
                    for (;;) {
                        const unsigned long long pmask = __ballot(pend);
                        if (!pmask) break;
                        const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)cs_j, __ffsll((long long)pmask) - 1);
                        const uint4 um = uni4(ix.blk_meta[blk]);
                        const uint32_t un = um.w & 0xff, umd = (um.w >> 8) & 0xff;
                        uint32_t a0, a1;
                        decode_doc_ids(blob + 8ull * um.z, umd, un, um.x, lane, a0, a1);
                        __builtin_amdgcn_wave_barrier();
                        *reinterpret_cast<uint2 *>(&scr[2 * lane]) = make_uint2(2 * lane < un ? a0 : NONE32, 2 * lane + 1 < un ? a1 : NONE32);
                        __builtin_amdgcn_wave_barrier();
                        if (pend && cs_j == blk) {
                            uint32_t p = 0;
#pragma unroll
                            for (int sft = 64; sft > 0; sft >>= 1)
                                if (scr[p + sft - 1] < cs_d) p += sft;
                            if (scr[p] == cs_d) cs_x = p;  // (the index: stage 2 leaves it alone)
                            else cs_fl = 0;
                            pend = false;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                cs_stage = 2;
            } else if (cs_stage == 2) {
                // ---- stage 2: the posting's index; its tf / fieldnorm word
                if ((cs_fl & 5u) == 5u) {
                    const uint32_t r = cs_d - cs_x, base = cs_fl >> 8;
                    uint32_t idx = NONE32;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if ((cs_v[u] & 0xffffu) == r) idx = base + 2 * u;
                        if ((cs_v[u] >> 16) == r) idx = base + 2 * u + 1;
                    }
                    cs_x = idx;
                    if (idx == NONE32) cs_fl = 0;
                }
                cs_fl &= 0xffu;
                if (cs_fl & 4u) {
                    if (cs_fl & 2u) {
                        cs_v[0] = ix.post_tfn[64ull * cs_j + (cs_x >> 1)];
                    } else {  // tf fields wider than 7 bits / tails: the generic decode of the one field (rare)
                        const TeamArgsP cg = cold_args();
                        const uint4 mm = ix.blk_meta[cs_j];
                        const uint32_t nj = mm.w & 0xff, mdj = (mm.w >> 8) & 0xff, mtj = (mm.w >> 16) & 0xff;
                        const uint8_t *tbody = cg->ix.blob + 8ull * mm.z + ((payload_bytes(mdj, nj) + 7u) & ~7u);
                        const FieldAddr fa = field_addr(mtj, nj, cs_x);
                        const uint32_t flo = *reinterpret_cast<const uint32_t *>(tbody + fa.off0);
                        const uint32_t fhi = *reinterpret_cast<const uint32_t *>(tbody + fa.off1);
                        cs_v[0] = field_val(flo, fhi, fa);
                        cs_v[1] = cg->ix.post_fn[128ull * cs_j + cs_x];
                    }
                }
                cs_stage = 3;
            } else if (cs_stage == 3) {
                // ---- stage 3: s1 of the posting's fieldnorm
                if (cs_fl & 4u) {
                    if (cs_fl & 2u) {
                        const uint32_t sh = (cs_x & 1u) * 8u;
                        cs_v[1] = (cs_v[0] >> (16u + sh)) & 0xffu;
                        cs_v[0] = (cs_v[0] >> sh) & 0xffu;
                    }
                    const double s1v = ix.s1[cs_v[1]];
                    cs_v[2] = (uint32_t)__double2loint(s1v);
                    cs_v[3] = (uint32_t)__double2hiint(s1v);
                }
                cs_stage = 4;
            } else {
                // ---- stage 4: Cache::evaluate (bm25.rs:355-358); the document's terms summed in ascending key order
                // (absent terms add 0.0, exact); the offer
                const double s0 = __shfl(r_s0, (int)cs_t);
                double c = 0.0;
                if (cs_fl & 4u) {
                    const double tfd = (double)cs_v[0];
                    c = (tfd * s0) / (tfd + __hiloint2double((int)cs_v[3], (int)cs_v[2]));
                }
                double *cs = reinterpret_cast<double *>(scr);
                __builtin_amdgcn_wave_barrier();
                cs[lane] = c;
                __builtin_amdgcn_wave_barrier();
                double acc = 0.0;
                const bool leader = cs_ci < cs_n && cs_t == 0;
                if (leader)
                    for (uint32_t u = 0; u < m; ++u) acc += cs[lane + u];
                __builtin_amdgcn_wave_barrier();
                {   // the offer: whole documents to this wave's list
                    const unsigned long long th = TM_THETA_NOW();
                    bool has = leader && (unsigned long long)__double_as_longlong(acc) >= th &&
                               (rtop.cnt < k || better(acc, cs_d, rtop.kth_s, rtop.kth_d));
                    if (__ballot(has)) {
                        rtop.template offer<true>(has, acc, cs_d, k, lane);
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
                    }
                }
                q_head += cs_n;
                if (cs_from == 0) q0 -= cs_n;
                else if (cs_from == 1) q1 -= cs_n;
                else q2 -= cs_n;
                cs_stage = 0;
            }
                    } while (false);
        }
        if (failed && lane == 0) S.fail = 1;
        __syncthreads();  // (every wave's verdict is in)
        const bool item_fails = uni(S.fail) != 0u;

        // ---- item result: one list per wave (merge_kernel merges them and drops a document scored by two waves)
        const uint32_t n = item_fails ? 0u : rtop.cnt;
        const TeamArgsP ce = cold_args();
        const size_t lst = (size_t)item * ce->bt.lpi + wave;
        double *res_score = ce->bt.res_score;
        uint32_t *res_doc = ce->bt.res_doc;
#pragma unroll
        for (int r = 0; r < RK; ++r)
            if (r * 64 + lane < n) {
                res_score[lst * k + r * 64 + lane] = rtop.score[r];
                res_doc[lst * k + r * 64 + lane] = rtop.doc[r];
            }
        if (lane == 0) {
            ce->bt.res_cnt[lst] = n;
            if (wave == 0 && item_fails) ce->bt.item_failed[item] = 1u;
        }
    }
}
#undef TM_THETA_NOW
#undef TM_ISSUE
#undef TM_AFTER_GROUPS
