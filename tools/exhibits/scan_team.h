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
//   completion  a candidate document is scored from scratch, whoever found it, ONCE PER ITEM and in bulk: lanes = (candidate,
//               term), four batches of 64 lanes interleaved in straight-line code.  The term's block that may hold the
//               document comes from the term's bucket locator (blk_loc: first block whose last document reaches the bucket,
//               buckets of about one block span; Cursor::seek_block, search.rs:412-431, in two round trips instead of a
//               bisection), the posting from a two-level search of the block's 16-bit ids (blk_piv: every 16th id, then 32
//               bytes of the plane), its tf / fieldnorm from post_tfn, then Cache::evaluate (bm25.rs:355-358) and the sum in
//               ascending key order (evaluate.rs:43-72) -> the wave's register top-k.  Six dependent round trips to memory:
//               done per window on the spot they doubled the kernel's time (every wave of a team waits for the slowest), as
//               a pipeline advanced between the groups of marks they cost the compiler its count of the loads in flight
//               (every wait became a wait for everything).  In bulk, four batches at a time, the chain costs 1.5 round
//               trips per batch and nobody waits for it but the wave itself.  A document found twice (three lists; a cold
//               posting that is also a second arrival; two waves) is scored twice to the same bits: RegTopK and
//               merge_kernel drop the repetition.  Lookups the fast path cannot serve (more than four blocks in a bucket,
//               blocks without a plane or tf word) are redone one by one, from the generic decode.
//   life cycle  of the bitmap, the only thing the team synchronises for: every mark of window n before it is wiped, the
//               wipe before the marks of n + 1.  Two LDS counters, arrive early / wait late, no s_barrier inside an item.
//
// A wave's candidates go to its list in global memory (bt.team_cand); an item whose lists intersect so densely that a
// list overflows is handed to scan_many_kernel (item_failed), as the round-3 kernel did.
//
// Threshold: theta0 = the largest, over the query's terms, of the k-th largest block maximum of the term
// (term_kth_ub, derived at index creation; every block maximum is the score of a posting of its block, documents of one
// term are distinct, and a document's score is at least any of its postings') -- a lower bound of the final k-th
// score before the first posting is read; then the waves' k-th scores through LDS and the query's 64-bit atomicMax
// word.  Filtering on score < threshold is exact: ties are kept.

constexpr int TM_TQ = 4;         // candidate blocks per term and round (16 terms x 4 = the 64 lanes)
constexpr int TM_G = 4;          // blocks per group of the branch-free path
constexpr int TM_KTH = 9;        // term_kth_ub entries per term: the 2^i-th largest block maximum, i = 0..8
constexpr int TM_IL = 4;         // batches of the bulk completion in flight together
constexpr uint32_t TM_CAND = 8192;  // candidate documents a wave's list holds (bt.team_cand)
constexpr uint32_t TM_FLUSH = 2048; // ... completed at the next round's start when it holds more than this

template <int TEAM>
struct TeamLds {
    uint32_t bm[TEAM * 2048];
    uint32_t scr[TEAM][128];       // per wave: ids of a block decoded for a lookup / the contributions of a batch
    unsigned long long theta;      // bits of a lower bound of the query's k-th best score
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

// (macros, not lambdas: with many by-reference closures in one large function the optimiser leaves every captured local --
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

typedef KernArgsP TeamArgsP;  // (device_types.h: the cold arguments read from the kernarg segment where they are used)

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
    uint32_t *const scr = S.scr[wave];
    uint32_t *const gl = bt.team_cand + (size_t)(blockIdx.x * TEAM + wave) * TM_CAND;  // this wave's candidate list
    const uint32_t slot_t = lane / TM_TQ, off = lane % TM_TQ;
    const float inv_docs = 1.0f / (float)ix.n_docs;
    uint32_t kidx = 0;  // term_kth_ub entry: the smallest 2^i >= k
    while ((1u << kidx) < k) ++kidx;

    for (uint32_t i = tid; i < WORDS; i += TEAM * 64) S.bm[i] = 0;
    if (tid < 2) S.sync[tid] = 0;
    uint32_t epoch = 0;  // windows this workgroup has finished: the counters take TEAM arrivals per window

    for (;;) {
        __syncthreads();  // previous item: every wave is past its last wait
        if (tid == 0) {
            const TeamArgsP ca = cold_args();
            const uint32_t drawn = atomicAdd(&ca->bt.work_ctr[0], 1u);
            S.item = drawn < n_items ? ca->bt.item_order[drawn] : NONE32;  // plan_kernel's order: longest first
            S.theta = 0;
            S.fail = 0;
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
        uint32_t r_b0 = 0, r_b1 = 0, r_df = 0, r_loc = 0, r_sh = 0;
        double r_s0 = 0.0, kth = 0.0;
        if (act) {
            r_b0 = ca->ix.term_first_block[term];
            r_b1 = ca->ix.term_first_block[term + 1];
            r_s0 = ca->ix.term_s0[term];
            r_df = ca->ix.term_df[term];
            const uint2 tl = ca->ix.term_loc[term];
            r_loc = tl.x;
            r_sh = tl.y;
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

        uint32_t cnt = 0;     // candidates in gl
        bool failed = false;  // ... more than it holds
        // ---- the first round's candidates
        uint32_t j = cur + (uint32_t)TEAM * off;
        bool valid = slot_act && j < end;
        uint4 meta = make_uint4(NONE32, 0, 0, 0);
        double ub = 0.0;
        if (valid) {
            meta = ix.blk_meta[j];
            ub = ix.blk_ub[j];
        }

        RegTopK<RK> rtop;
        rtop.init();
        unsigned long long published = 0;
        // (one pass more than there are windows: the last pass only completes what is left in the list)
        for (uint32_t tlo = lo;;) {
            const bool last_pass = tlo >= hi;
            const uint32_t thi = last_pass ? tlo : hi - tlo > W ? tlo + W : hi, span = thi - tlo;
            unsigned long long poll = 0;  // the query's threshold as the other workgroups see it: asked for now, used at the window's end
            if (wave == 0 && !last_pass) poll = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {  // rounds: up to TM_TQ blocks per term each
                if (cnt > TM_CAND) {
                    failed = true;  // one round made more candidates than the list holds: the item is scan_many_kernel's
                    cnt = 0;
                }
                if (cnt && (last_pass || cnt > TM_FLUSH) && !(bt.team_dbg & 1u)) {
                    // ---- bulk completion of the wave's candidates: TM_IL batches of 64 / m candidates x m terms at a time, every
                    // stage for all of them before the next stage (straight-line code: the loads of the four batches are in flight together)
                    const uint32_t inv_m = (65536u + m - 1u) / m;  // p / m == (p * inv_m) >> 16 for p < 4096
                    const uint32_t per_batch = 64u / m;
                    const uint32_t ci = (lane * inv_m) >> 16, t = lane - ci * m;
                    const uint32_t tb1 = (uint32_t)__shfl((int)r_b1, (int)t), tb0 = (uint32_t)__shfl((int)r_b0, (int)t);
                    const uint32_t tloc = (uint32_t)__shfl((int)r_loc, (int)t), tsh = (uint32_t)__shfl((int)r_sh, (int)t);
                    const double ts0 = __shfl(r_s0, (int)t);
                    const TeamArgsP cb = cold_args();
                    const uint32_t *blk_loc = cb->ix.blk_loc, *blk_max_doc = cb->ix.blk_max_doc;
                    for (uint32_t c0 = 0; c0 < cnt; c0 += TM_IL * per_batch) {
                        uint32_t d[TM_IL], jb[TM_IL], fl[TM_IL], xx[TM_IL];  // fl: 1 plane, 2 tfn, 4 block found, 8 redo from the generic decode
                        uint32_t v[TM_IL][8];
                        bool task[TM_IL];
                        // ---- A: the candidates; the locator's bucket
        #pragma unroll
                        for (int u = 0; u < TM_IL; ++u) {
                            task[u] = ci < per_batch && c0 + u * per_batch + ci < cnt;
                            d[u] = task[u] ? gl[c0 + u * per_batch + ci] : 0u;
                            fl[u] = 0;
                            jb[u] = tb1;
                            if (task[u]) {
                                const uint32_t b = tloc + (d[u] >> tsh);
                                v[u][0] = blk_loc[b];
                                v[u][1] = blk_loc[b + 1];
                            }
                        }
                        // ---- B: the last documents of the bucket's first four blocks
        #pragma unroll
                        for (int u = 0; u < TM_IL; ++u) {
                            if (task[u]) {
                                const uint32_t ja = v[u][0], last = tb1 - 1u;
                                xx[u] = v[u][1];  // the bucket's last candidate block (tb1: none)
                                jb[u] = ja;
        #pragma unroll
                                for (int i = 0; i < 4; ++i) v[u][4 + i] = blk_max_doc[min(ja + (uint32_t)i, last)];
                            }
                        }
                        // ---- C: the block; its first document, flags and pivots
        #pragma unroll
                        for (int u = 0; u < TM_IL; ++u) {
                            if (task[u]) {
                                const uint32_t ja = jb[u], je = xx[u];
                                uint32_t jj = tb1;
        #pragma unroll
                                for (int i = 3; i >= 0; --i)
                                    if (ja + (uint32_t)i < tb1 && v[u][4 + i] >= d[u]) jj = ja + (uint32_t)i;
                                if (jj == tb1 && ja + 4u <= je && ja + 4u < tb1) fl[u] = 8;  // more than four blocks in the bucket: redone below
                                jb[u] = jj;
                                if (jj < tb1) {
                                    const uint4 mm = ix.blk_meta[jj];
                                    const uint4 pv = ix.blk_piv[jj];
                                    v[u][0] = mm.x;
                                    v[u][1] = mm.y;
                                    v[u][2] = mm.w;
                                    v[u][4] = pv.x;
                                    v[u][5] = pv.y;
                                    v[u][6] = pv.z;
                                    v[u][7] = pv.w;
                                    fl[u] = 4;
                                }
                            }
                        }
                        // ---- D: the 16 ids that can hold the document
        #pragma unroll
                        for (int u = 0; u < TM_IL; ++u) {
                            if (fl[u] & 4u) {
                                const uint32_t mn = v[u][0];
                                if (mn > d[u]) {
                                    fl[u] = 0;  // the document lies between two blocks
                                } else if (!rel16_block(mn, v[u][1], v[u][2])) {
                                    fl[u] = 8;
                                } else {
                                    const uint32_t r = d[u] - mn;
                                    uint32_t seg = 0;
        #pragma unroll
                                    for (int i = 4; i < 8; ++i) seg += ((v[u][i] & 0xffffu) < r ? 1u : 0u) + ((v[u][i] >> 16) < r ? 1u : 0u);
                                    if (seg > 7u) {
                                        fl[u] = 0;
                                    } else if (!tfn_block(v[u][2])) {
                                        fl[u] = 8;
                                    } else {
                                        xx[u] = r | (16u * seg) << 16;
                                        const uint4 *pp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(ix.post_rel16) + 128ull * jb[u] + 16u * seg);
                                        const uint4 a = pp[0], b = pp[1];
                                        v[u][0] = a.x;
                                        v[u][1] = a.y;
                                        v[u][2] = a.z;
                                        v[u][3] = a.w;
                                        v[u][4] = b.x;
                                        v[u][5] = b.y;
                                        v[u][6] = b.z;
                                        v[u][7] = b.w;
                                    }
                                }
                            }
                        }
                        // ---- E: the posting's index; its tf / fieldnorm word
        #pragma unroll
                        for (int u = 0; u < TM_IL; ++u) {
                            if (fl[u] & 4u) {
                                const uint32_t r = xx[u] & 0xffffu, base = xx[u] >> 16;
                                uint32_t idx = NONE32;
        #pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    if ((v[u][i] & 0xffffu) == r) idx = base + 2 * i;
                                    if ((v[u][i] >> 16) == r) idx = base + 2 * i + 1;
                                }
                                if (idx == NONE32) {
                                    fl[u] = 0;
                                } else {
                                    xx[u] = idx;
                                    v[u][0] = ix.post_tfn[64ull * jb[u] + (idx >> 1)];
                                }
                            }
                        }
                        // ---- F: s1 of the posting's fieldnorm
        #pragma unroll
                        for (int u = 0; u < TM_IL; ++u) {
                            if (fl[u] & 4u) {
                                const uint32_t sh = (xx[u] & 1u) * 8u;
                                const uint32_t fn = (v[u][0] >> (16u + sh)) & 0xffu;
                                v[u][0] = (v[u][0] >> sh) & 0xffu;
                                const double s1v = ix.s1[fn];
                                v[u][2] = (uint32_t)__double2loint(s1v);
                                v[u][3] = (uint32_t)__double2hiint(s1v);
                            }
                        }
                        // ---- G: Cache::evaluate (bm25.rs:355-358); the lookups the fast path could not serve; the document's terms summed
                        // in ascending key order (absent terms add 0.0, exact); the offer
        #pragma unroll
                        for (int u = 0; u < TM_IL; ++u) {
                            if (c0 + u * per_batch >= cnt) break;
                            double c = 0.0;
                            if (fl[u] & 4u) {
                                const double tfd = (double)v[u][0];
                                c = (tfd * ts0) / (tfd + __hiloint2double((int)v[u][3], (int)v[u][2]));
                            }
                            if (__ballot((fl[u] & 8u) != 0)) {  // from the generic decode, one block at a time (rare)
                                const TeamArgsP cg = cold_args();
                                uint32_t jj = tb1;
                                uint4 mm = make_uint4(0, 0, 0, 0);
                                bool pend = false;
                                if (fl[u] & 8u) {
                                    jj = tm_first_block_ge(blk_max_doc, tb0, tb1, d[u], inv_docs);
                                    if (jj < tb1) {
                                        mm = ix.blk_meta[jj];
                                        pend = mm.x <= d[u];
                                    }
                                }
                                uint32_t idx = NONE32;
                                for (;;) {
                                    const unsigned long long pmask = __ballot(pend);
                                    if (!pmask) break;
                                    const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)jj, __ffsll((long long)pmask) - 1);
                                    const uint4 um = uni4(ix.blk_meta[blk]);
                                    const uint32_t un = um.w & 0xff, umd = (um.w >> 8) & 0xff;
                                    uint32_t a0, a1;
                                    decode_doc_ids(cg->ix.blob + 8ull * um.z, umd, un, um.x, lane, a0, a1);
                                    __builtin_amdgcn_wave_barrier();
                                    *reinterpret_cast<uint2 *>(&scr[2 * lane]) = make_uint2(2 * lane < un ? a0 : NONE32, 2 * lane + 1 < un ? a1 : NONE32);
                                    __builtin_amdgcn_wave_barrier();
                                    if (pend && jj == blk) {
                                        uint32_t p = 0;
        #pragma unroll
                                        for (int sft = 64; sft > 0; sft >>= 1)
                                            if (scr[p + sft - 1] < d[u]) p += sft;
                                        if (scr[p] == d[u]) idx = p;
                                        pend = false;
                                    }
                                    __builtin_amdgcn_wave_barrier();
                                }
                                if (idx != NONE32) {
                                    const uint32_t nj = mm.w & 0xff, mdj = (mm.w >> 8) & 0xff, mtj = (mm.w >> 16) & 0xff;
                                    const uint8_t *tbody = cg->ix.blob + 8ull * mm.z + ((payload_bytes(mdj, nj) + 7u) & ~7u);
                                    const FieldAddr fa = field_addr(mtj, nj, idx);
                                    const uint32_t flo = *reinterpret_cast<const uint32_t *>(tbody + fa.off0);
                                    const uint32_t fhi = *reinterpret_cast<const uint32_t *>(tbody + fa.off1);
                                    const double tfd = (double)field_val(flo, fhi, fa);
                                    c = (tfd * ts0) / (tfd + ix.s1[cg->ix.post_fn[128ull * jj + idx]]);
                                }
                            }
                            double *cs = reinterpret_cast<double *>(scr);
                            __builtin_amdgcn_wave_barrier();
                            cs[lane] = c;
                            __builtin_amdgcn_wave_barrier();
                            double acc = 0.0;
                            const bool leader = task[u] && t == 0;
                            if (leader)
                                for (uint32_t i = 0; i < m; ++i) acc += cs[lane + i];
                            __builtin_amdgcn_wave_barrier();
                            {   // the offer: whole documents to this wave's list
                                const unsigned long long th = TM_THETA_NOW();
                                const bool has = leader && (unsigned long long)__double_as_longlong(acc) >= th &&
                                                 (rtop.cnt < k || better(acc, d[u], rtop.kth_s, rtop.kth_d));
                                if (__ballot(has)) {
                                    rtop.template offer<true>(has, acc, d[u], k, lane);
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
                        }
                    }
                    cnt = 0;
                }
                if (bt.team_dbg & 1u) cnt = 0;
                if (last_pass) break;
                const double thd = __longlong_as_double((long long)TM_THETA_NOW());
                const bool in_win = valid && meta.x < thi;
                const bool plane = shift == 0u && rel16_block(meta.x, meta.y, meta.w);
                const bool cold = ub >= thd;  // search.rs:203 per block (blk_ub carries its margin)
                unsigned long long m_fast = __ballot(in_win && plane && !cold);
                const unsigned long long m_cold = __ballot(in_win && cold);
                unsigned long long m_slow = __ballot(in_win && !(plane && !cold));
                const unsigned long long done = __ballot(in_win && meta.y < thi);
                const uint32_t ndone = (uint32_t)__popcll((done >> (slot_t * TM_TQ)) & ((1ull << TM_TQ) - 1ull));
                const bool again = __ballot(slot_act && ndone == (uint32_t)TM_TQ) != 0ull;
                const uint32_t v_delta = meta.x - tlo, v_blk = in_win ? j : 0u;  // (a padded group entry reads lane 0's block: a valid one)
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

                // ---- groups of TM_G blocks with a plane word: branch-free, the next group's words in flight
                const uint32_t ngroups = ((uint32_t)__popcll(m_fast) + TM_G - 1) / TM_G;
                if (ngroups) {
                    uint32_t w[TM_G], dl[TM_G];
                    TM_ISSUE();
                    for (uint32_t g = 0; g < ngroups; ++g) {
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
                        if (any) {  // second arrivals -> the candidate list
#pragma unroll
                            for (int p = 0; p < 2 * TM_G; ++p) {
                                if (dm[p]) {
                                    if ((o[p] & mk[p]) != 0) {
                                        const uint32_t at = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm[p] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm[p], 0u));
                                        if (at < TM_CAND) gl[at] = tlo + x[p];
                                    }
                                    cnt += (uint32_t)__popcll(dm[p]);
                                }
                            }
                        }
                    }
                }

                // ---- general path, one block at a time: blocks without a plane word, windows of 2^shift documents per bit,
                // blocks whose upper bound reaches the threshold (every posting that could reach it alone is a candidate)
                while (m_slow) {
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
                    const unsigned long long e0 = __ballot(f0), e1 = __ballot(f1);
                    if (f0) {
                        const uint32_t at = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(e0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)e0, 0u));
                        if (at < TM_CAND) gl[at] = d0;
                    }
                    cnt += (uint32_t)__popcll(e0);
                    if (f1) {
                        const uint32_t at = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(e1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)e1, 0u));
                        if (at < TM_CAND) gl[at] = d1;
                    }
                    cnt += (uint32_t)__popcll(e1);
                }
                if (!again) break;
            }

            if (last_pass) break;
            // ---- the window's marks of this wave are in: the bitmap's life cycle
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(&S.sync[0], 1u);
            if (wave == 0 && poll > TM_THETA_NOW() && lane == 0) atomicMax(&S.theta, poll);
            {
                const uint32_t target = (epoch + 1u) * TEAM;
                while (__hip_atomic_load(&S.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                for (uint32_t i = tid; i < WORDS / 4; i += TEAM * 64) reinterpret_cast<uint4 *>(S.bm)[i] = make_uint4(0, 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&S.sync[1], 1u);
                while (__hip_atomic_load(&S.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
            }
            ++epoch;
            tlo = thi;
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
            if (wave == 0 && item_fails) {
                ce->bt.item_failed[item] = 1u;
                *ce->bt.fail_any = 1u;
            }
        }
    }
}
#undef TM_THETA_NOW
#undef TM_ISSUE
