// scan_dense.h -- scan_dense_kernel: queries with many postings per document (Zipf head terms; BASELINE config
// C5), <= D_T indexed terms, k <= REG_K.  Part of libvbm25's single device translation unit: included by
// search.hip inside namespace vbm25, after scan_range.h (whose helpers it uses).
//
// One 8-wave workgroup per work item (query x doc range), persistent, items from bt.work_ctr[1].  The item is
// cut into WINDOWS of <= D_W consecutive documents with one 16-bit FIXED-POINT accumulator per document in LDS (two per word)
// (integer LDS atomics run at full rate on gfx950; float LDS atomics measured three times slower than the
// whole rest of a window).  A window is
//
//   P0  enumerate: the blocks of every term that start below the window's end (one lane per block: 16-byte
//       metadata + block upper bound, requested a window ahead), written as TASKS straight into the window's task
//       list: a term takes its place with one LDS atomic -- from the bottom of the list upwards, the HEAD terms (below)
//       from the top downwards -- so nobody waits for the other terms' counts.  Wave 0 polls the threshold and takes
//       the MaxScore split the NEXT window is laid out for.  One barrier.
//   P1  every term but the heads (search.rs:153-169: the essential terms of the MaxScore split on token upper bounds,
//       and the non-essential ones that are not dense): every task fetched -- the lane's words of post_rel16 and
//       post_tfn: two ids, two term frequencies, two fieldnorms (blocks without plane words: generic decode of the
//       blob) -- and an UPPER BOUND of each posting's score (f32 rounded up, scaled to an integer, + 1) added to its
//       document's accumulator, in any order: no barrier between terms, all eight waves busy, groups of four blocks
//       per wave, the next group issued before the current one is added up.
//   P2  the HEAD terms (non-essential and df >= N / 2), after a barrier.  A block is fetched only if some document
//       of its span can still reach the threshold: max accumulator over the span + the bounds of the other heads +
//       the block's own upper bound (search.rs:177-203: the block-max test).  Every other block is SKIPPED: no id,
//       tf or fieldnorm byte of it is read.  The test stands on its own for ANY set of tested terms, so the split a
//       window is laid out for may be one window old (the threshold only rises: the set only grows).
//   P3  after a barrier (every sum of the window complete): candidates -- documents whose accumulated bound reaches
//       the threshold go to the wave's candidate buffer (document, bound) and -- by the LOWER bound of their score
//       that the accumulator also implies -- into the query's 256-bucket histogram (bt.hist, shared by all items of
//       the query, as in scan_range.h): k documents in buckets >= b put the final k-th score at or above the lower
//       edge of b.  The threshold therefore rises from approximate sums alone, across items, without any exact
//       score.  Accumulators wiped in the same pass; no barrier behind it (nobody adds to them before the next
//       window's P0 barrier).
// Three barriers per window (two when no term is a head); rounds 2-4 took six (two in P0, one behind P3, two around
// a flush).
//
// FLUSH (some wave's buffer half full: every wave, before the next window's P3; a wave alone when a bucket no longer
// fits; always at the item's end), after dropping the entries the threshold has overtaken:
// the surviving candidates are re-scored EXACTLY -- block located by interpolation + gallop + bisection of
// blk_max_doc, block upper bounds first (search.rs:177-203), then the block decoded by the wave and the one
// posting's f64 Cache::evaluate (bm25.rs:355-358) taken, sum in ascending key order (evaluate.rs:43-72) -> the
// wave's register top-k -> the query's shared threshold.  Most candidates of the warm-up are never re-scored.
//
// The integer sums only SELECT; every score that is compared, kept or returned is the exact f64 sum, so results
// are bit-identical to the other kernels'.  Bounds: s0 is rounded up to f32 and carries a factor 1 + 2^-19
// (five f32 roundings + v_rcp_f32's 1 ulp < 2^-21 relative), s1 is rounded down; the scale is a power of two with
// scale x (sum of the token upper bounds) < 2^15 (the accumulators are 16 bits wide); truncation to an integer is covered by the + 1; the integer sum is
// exact.  So  acc >= scale x score  and  score >= (acc - m) / (scale (1 + 2^-18)).
// oracle/dense_model.inc is a scalar CPU model of this scheme (tests/test_dense_model.py).
//
// The first window of a query whose threshold is still 0 is 256 documents wide and the width doubles from
// there: the number of candidates per window stays near k ln 2 while the threshold warms up.

constexpr int DNW = 8;
static_assert(DNW <= RNW, "one result list per wave: bt.lpi is scan_range_kernel's");
constexpr int DWG = DNW * 64;
constexpr int D_W = 16384;               // documents per window (at most; an item of very dense terms takes narrower ones)
constexpr int D_W0 = 256;                // first window while the threshold is 0
constexpr int D_T = 16;                  // indexed terms per query
constexpr int D_KMAX = 256;              // largest k (register top-k of up to four rows per wave)
constexpr int D_SEG = 128;               // blocks of one term per window: two chunks of 64 lanes (a full block spans >= 128 documents)
constexpr int D_TCAP = 1024;             // blocks of all terms per window; an item's window width is chosen for 80 % of it
constexpr int D_WCB = 128;               // candidate buffer entries per wave
#ifndef D_UN_V
#define D_UN_V 4
#endif
constexpr int D_UN = D_UN_V;                  // tasks per wave in flight
constexpr uint32_t D_MAX_RESOLVED = 16384;  // exact re-scorings per item; beyond (masses of equal scores): scan_many_kernel
constexpr uint32_t D_SPAN_TEST = 512;    // widest block span the skip test reads (8 accumulators per lane)
constexpr uint32_t D_GRID = 512;         // persistent workgroups: 256 CUs x 2
constexpr uint32_t D_TARGET_ITEMS = 12288;   // items of a batch of dense queries: the launch ends with its slowest workgroup, and an item's setup is 24 k cycles of
                                           // the 20 M an item of C5 takes at 4096 -- C5: 4096 items 72.9 ms, 8192 69.9, 12288 69.7, 24576 70.9 (search.hip caps a
                                           // query's share where the corpus is small)

struct DenseLds {
    uint32_t acc[D_W / 2];    // 16-bit fixed point, two documents per word: scale x (upper bound of the document's score) < 2^16
    uint32_t bmax[D_W / 64];  // largest accumulator of every 64 documents, kept current by the adds
    uint4 tmeta[D_TCAP];      // {min_doc, max_doc, off8, n | md << 8 | mt << 16 | wand_fn << 24}
    uint32_t tblk[D_TCAP];    // block index
    uint32_t tub[D_TCAP];     // block upper bound, scaled, rounded up
    uint8_t tterm[D_TCAP];
    uint32_t cdoc[DNW][D_WCB], cval[DNW][D_WCB];  // per wave: candidate buffer (document, accumulator)
    uint32_t scr[DNW][128];   // per wave: ids of the block being looked into
    double s1[256];
    float s1f[256];           // rounded down
    double t_s0[D_T], t_ub[D_T], t_cum[D_T + 1];
    float t_s0i[D_T];         // scale x s0, rounded up, x (1 + 2^-19)
    uint32_t t_cur[D_T], t_end[D_T], t_b0[D_T];
    uint8_t t_rank[D_T], t_ord[D_T];
    uint8_t t_cls[D_T];       // document-frequency class: 2 = df >= N / 2, 1 = df >= N / 8, 0 = rarer
    uint32_t t_rem[D_T];      // non-essential term: scaled bounds of the OTHER terms that are incomplete during its phase
    double scale, hscale;
    unsigned long long theta; // bits of a lower bound of the query's k-th best score
    uint32_t item, m, fail, p_ne, h_ne, resolved, wmax;
    // by window parity: tasks of the window laid out so far from the bottom of the task list (nlo: every term but the heads) and
    // from its top (nhi: the head terms), and "some wave's candidate buffer is half full" (cfl, raised in P3, acted on by
    // everybody after the next window's barrier).  The entry of the other parity is zeroed while this one is in use.
    uint32_t nlo[2], nhi[2], cfl[2];
    uint32_t scratch[64];
};

// Register budget: at 128 registers (two workgroups per CU) the four-row instantiation spills 53 registers, and a build of
// this kernel under that pressure returned incomplete lists / faulted on the codec corner-case index (round 2).  The same
// source compiled for 256 registers is correct in every run, so is the present source at 128 (tools/dense_stress.py, 360
// runs), and no assertion of the -DVBM25_CHECK build ever fires: the defect follows the compiler's spill code, not an
// index or a race of this file.  k > 128 is therefore compiled without register pressure (one workgroup per
// CU: still an order of magnitude faster than the exhaustive kernel); k <= 128 keeps two workgroups per CU and is covered
// by the repetition test of tests/test_gpu_dense.py.
template <int KMAX>
__global__ void __launch_bounds__(DWG, KMAX > 128 ? 2 : 4) scan_dense_kernel(DevIndex ix, DevBatch bt) {
    static_assert(KMAX <= D_KMAX, "register top-k of at most four rows");
    constexpr int RK = KMAX / 64;
    __shared__ DenseLds S;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t k = bt.k;
    const uint32_t n_items = *cold_args()->bt.n_items;
    for (uint32_t i = tid; i < 256; i += DWG) {
        const double v = ix.s1[i];
        S.s1[i] = v;
        S.s1f[i] = __double2float_rd(v);
    }
    for (uint32_t i = tid; i < (uint32_t)D_W / 2; i += DWG) S.acc[i] = 0u;
    if (tid < (uint32_t)D_W / 64) S.bmax[tid] = 0u;
#ifdef VBM25_PROFILE
    // per wave: 0 windows, 1 P0, 2 P1, 3 P2, 4 wait before P3, 5 P3, 6 flush: filter, 7 flush: exact re-scoring, 8 item setup,
    // 9 window loops, 10 candidates buffered, 11 candidates re-scored, 12 items, 13 tasks fetched, 14 tasks skipped, 15 lifetime
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif

    for (;;) {
        __syncthreads();  // previous item fully done with LDS
        if (tid == 0) {
            S.item = atomicAdd(&cold_args()->bt.work_ctr[1], 1u);
            S.fail = 0;
            S.resolved = 0;
            S.h_ne = 0;
            S.nlo[0] = S.nlo[1] = S.nhi[0] = S.nhi[1] = S.cfl[0] = S.cfl[1] = 0;
        }
        __syncthreads();
        const uint32_t item = uni(S.item);
        if (item >= n_items) break;
        const Item it = cold_args()->bt.items[item];
        if (!(it.m & ITEM_DENSE) || (it.m & ~ITEM_DENSE) > (uint32_t)D_T) continue;  // the other kernels'
        const uint32_t q = uni(it.q), lo = uni(it.doc_lo), hi = uni(it.doc_hi);
        uint32_t *hrow = bt.hist + (size_t)q * CUR_HB;
        PROF_T(t_item);

        // ---- threshold poll (wave 0): the query's published k-th score and the histogram of the candidates' lower bounds
        unsigned long long pg = 0;
        uint32_t pc[4] = {0, 0, 0, 0};
        auto poll_request = [&]() {
            pg = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 4; ++i) pc[i] = __hip_atomic_load(&hrow[4 * lane + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto poll_consume = [&]() {
            const unsigned long long seen = ((unsigned long long)uni((uint32_t)(pg >> 32)) << 32) | uni((uint32_t)pg);
            unsigned long long th = seen;
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
                // a document lands in bucket b only if (a lower bound of) its score x hscale >= b (up to one rounding)
                const double edge = ((double)b / S.hscale) * (1.0 - 1e-12);
                const unsigned long long eb2 = (unsigned long long)__double_as_longlong(edge);
                if (eb2 > th) th = eb2;
            }
            if (lane == 0) {
                atomicMax(&S.theta, th);
                if (th > seen) atomicMax(&bt.theta[q], th);  // (for merge_kernel too: entries below the threshold need no merging)
            }
        };

        // ---- item setup (wave 0, lane t = term t): block ranges, first block at or after lo, bounds, order, scale
        if (wave == 0) {
            poll_request();
            const KernArgsP ca = cold_args();  // (the item setup's pointers: not kept in registers over the window loop)
            uint32_t m = 0, term = NONE32;
            {
                const uint32_t qb = uni(ca->bt.q_off[q]), qe = uni(ca->bt.q_off[q + 1]);
                for (uint32_t base = qb; base < qe; base += 64) {  // compaction of the indexed terms (search.rs:59-61)
                    const uint32_t tt = base + lane < qe ? ca->bt.term_ids[base + lane] : NONE32;
                    const bool ok = tt < ix.n_terms;
                    const unsigned long long okm = __ballot(ok);
                    const uint32_t pos = m + __builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u));
                    if (ok && pos < 64u) S.scratch[pos] = tt;
                    m += (uint32_t)__popcll(okm);
                }
                __builtin_amdgcn_wave_barrier();
                m = min(uni(m), (uint32_t)D_T);  // (more than D_T: filtered above through it.m)
                if (lane < m) term = S.scratch[lane];
                __builtin_amdgcn_wave_barrier();
            }
            const bool act = lane < m;
            double s0 = 0.0, tub = 0.0, kth = 0.0;
            const double *kub = ca->ix.term_kth_ub;
            if (act && kub) {  // theta0: the term's 2^i-th largest block maximum, 2^i >= k (k documents of the term score that much)
                uint32_t kidx = 0;
                while ((1u << kidx) < k) ++kidx;
                kth = kub[(size_t)term * 9 + kidx];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) kth = fmax(kth, __shfl_xor(kth, o));
            const unsigned long long theta0 = (unsigned long long)__double_as_longlong(kth);
            if (act) {
                // (sl: the lane number, made opaque here -- the LDS addresses below are otherwise computed once at kernel
                // entry, for every wave, and kept in scratch over the whole launch)
                uint32_t sl = lane;
                asm volatile("" : "+v"(sl));
                const uint32_t b0 = ca->ix.term_first_block[term], b1 = ca->ix.term_first_block[term + 1];
                s0 = ca->ix.term_s0[term];
                const double wtf = (double)ca->ix.term_wand_tf[term];
                tub = ((wtf * s0) / (wtf + S.s1[ca->ix.term_wand_fn[term]])) * (1.0 + 1e-12);
                S.t_cur[sl] = r_first_block_ge(ix, b0, b1, lo);
                S.t_b0[sl] = b0;
                S.t_end[sl] = b1;
                S.t_s0[sl] = s0;
                S.t_ub[sl] = tub;
                const uint32_t df = ca->ix.term_df[term];
                S.t_cls[sl] = (uint8_t)(df >= ix.n_docs / 2 ? 2 : df >= ix.n_docs / 8 ? 1 : 0);
            }
            uint32_t rank = 0;  // position in ascending order of the token upper bounds
            double sums0 = 0.0;
            for (uint32_t t = 0; t < m; ++t) {
                const double ubt = readlane_f64(tub, t);
                if (act && (ubt < tub || (ubt == tub && t < lane))) ++rank;
                sums0 += readlane_f64(s0, t);
            }
            if (act) {
                S.t_rank[lane] = (uint8_t)rank;
                S.t_ord[rank] = (uint8_t)lane;
            }
            double cum = 0.0;
            {   // (a zero made here: the compiler otherwise keeps a 64-bit zero from kernel entry on -- in scratch memory)
                int z = 0;
                asm volatile("" : "+v"(z));
                if (lane == 0) S.t_cum[0] = __hiloint2double(z, z);
            }
            for (uint32_t pp = 0; pp < m; ++pp) {
                const uint32_t owner = (uint32_t)__ffsll((long long)__ballot(act && rank == pp)) - 1u;
                cum += readlane_f64(tub, owner);
                if (lane == 0) S.t_cum[pp + 1] = cum;
            }
            // power-of-two scale with scale x (sum of all token bounds) < 2^15: a document's sum (+ 1 per posting, + 2^-18
            // relative) stays below 2^16
            int e = 14 - ilogb(cum > 0.0 ? cum : 1.0);
            e = e > 60 ? 60 : (e < -60 ? -60 : e);
            const double scale = ldexp(1.0, e);
            if (act) S.t_s0i[lane] = (__double2float_ru(s0) * (1.0f + 1.0f / 524288.0f)) * (float)scale;
            // window width: the expected number of blocks per window (sum of df / 128 per document) within 80 % of the task
            // list, no term above 120 blocks (chunks of 64 lanes: 128)
            unsigned long long sumdf = 0;
            uint32_t maxdf = 1;
            for (uint32_t t = 0; t < m; ++t) {
                const uint32_t dft = (uint32_t)__builtin_amdgcn_readlane((int)(act ? ca->ix.term_df[term] : 0u), (int)t);
                sumdf += dft;
                maxdf = max(maxdf, dft);
            }
            unsigned long long wfit = (unsigned long long)(D_TCAP * 4 / 5) * 128ull * ix.n_docs / max(sumdf, 1ull);
            wfit = min(wfit, 120ull * 128ull * ix.n_docs / maxdf);
            const uint32_t wmax_item = (uint32_t)min((unsigned long long)D_W, max(wfit & ~1023ull, 1024ull));
            if (lane == 0) {
                S.m = m;
                S.wmax = wmax_item;
                S.scale = scale;
                S.hscale = (double)CUR_HB / sums0;  // score -> histogram bucket: linear in [0, sum of s0), as scan_range.h
                S.theta = theta0;  // a lower bound of the final k-th score before the first posting is read
                if (theta0) atomicMax(&bt.theta[q], theta0);
            }
            __builtin_amdgcn_wave_barrier();
            poll_consume();
        }
        __syncthreads();
        const uint32_t m = uni(S.m);
        const double scale = S.scale;
        // threshold -> fixed point, rounded down (an accumulator >= this may belong to a document at or above the threshold)
        auto theta_fix = [&](unsigned long long th) -> uint32_t {
            const double x = __longlong_as_double((long long)th) * scale;
            return x >= 4294967295.0 ? 0xffffffffu : (uint32_t)x;
        };

        RegTopK<RK> rtop;
        rtop.init();
        unsigned long long published = 0;
        auto theta_now = [&]() -> unsigned long long {
            const unsigned long long th = S.theta;
            return ((unsigned long long)uni((uint32_t)(th >> 32)) << 32) | uni((uint32_t)th);
        };
        auto offer = [&](bool has, double sc, uint32_t d) {
            const unsigned long long th = theta_now();
            has = has && (unsigned long long)__double_as_longlong(sc) >= th &&
                  (rtop.cnt < k || better(sc, d, rtop.kth_s, rtop.kth_d));
            if (!__ballot(has)) return;
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

        // ---- one task: decode the block, add the postings' upper bounds to their documents' accumulators
        struct Raw {
            uint32_t rel, tfn;  // the lane's words of post_rel16 and post_tfn: two postings
        };
        // bucket maxima are read by the skip test of P2 (largest accumulator of a span + the bounds the head terms can still add
        // >= threshold) and by P3 (>= threshold): a sum below bmax_floor = threshold - (all the head terms' bounds) can make neither
        // true, so only the sums at or above it are recorded -- with a warm threshold almost none, and the same-address LDS
        // atomics of 64 postings that share a few buckets go away
        uint32_t bmax_floor = 0;
        auto add_pair = [&](float s0i, uint32_t d0, uint32_t d1, uint32_t f0, uint32_t f1, uint32_t fn, bool in0, bool in1,
                            uint32_t wlo, uint32_t wspan) {
            VCHK(wspan <= (uint32_t)D_W, 10, wspan);
            const uint32_t x0 = d0 - wlo, x1 = d1 - wlo;
            const float tf0 = (float)f0, tf1 = (float)f1;
            const uint32_t p0 = (uint32_t)((tf0 * s0i) * __builtin_amdgcn_rcpf(tf0 + S.s1f[fn & 0xff])) + 1u;
            const uint32_t p1 = (uint32_t)((tf1 * s0i) * __builtin_amdgcn_rcpf(tf1 + S.s1f[fn >> 8])) + 1u;
            VCHK(!(in0 && x0 < wspan) || p0 < 32768u, 11, p0);
            VCHK(!(in1 && x1 < wspan) || p1 < 32768u, 11, p1);
            // the add that comes last in an accumulator's order sees the final sum: the bucket maximum is never below it
            // (no carry between the halves of a word: every document's sum stays below 2^16)
            if (in0 && x0 < wspan) {
                const uint32_t sum = ((atomicAdd(&S.acc[x0 >> 1], p0 << (16u * (x0 & 1u))) >> (16u * (x0 & 1u))) & 0xffffu) + p0;
                VCHK(sum < 65536u, 12, sum);
                if (sum >= bmax_floor) atomicMax(&S.bmax[x0 >> 6], sum);
            }
            if (in1 && x1 < wspan) {
                const uint32_t sum = ((atomicAdd(&S.acc[x1 >> 1], p1 << (16u * (x1 & 1u))) >> (16u * (x1 & 1u))) & 0xffffu) + p1;
                VCHK(sum < 65536u, 12, sum);
                if (sum >= bmax_floor) atomicMax(&S.bmax[x1 >> 6], sum);
            }
        };
        // byte-packed tail, raw or wide block, term frequencies of 8 bits and more: generic, synchronous decode from the blob
        // (rare: one call site)
        auto task_slow = [&](uint32_t e, uint32_t wlo, uint32_t wspan) {
            const uint4 c = uni4(S.tmeta[e]);
            const uint32_t j = uni(S.tblk[e]), t = uni((uint32_t)S.tterm[e]);
            const uint32_t n = c.w & 0xff, md = (c.w >> 8) & 0xff, mt = (c.w >> 16) & 0xff;
            const uint8_t *body = ix.blob + 8ull * c.z;
            uint32_t d0, d1, f0, f1;
            decode_doc_ids(body, md, n, c.x, lane, d0, d1);
            decode_fields(body + ((payload_bytes(md, n) + 7u) & ~7u), mt, n, lane, f0, f1);
            const uint32_t fn = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * j)[lane];
            add_pair(S.t_s0i[t], d0, d1, f0, f1, fn, 2 * lane < n, 2 * lane + 1 < n, wlo, wspan);
        };
        // A GROUP: up to D_UN tasks of this wave (entries e[0 .. nv) of the task list), all of them in flight together.
        // grp_issue reads the entries and starts the loads, grp_finish adds the postings up; P1 issues group n + 1
        // before it finishes group n, so that the round trip to memory and the entry reads hide behind the adds.
        static_assert(D_TCAP <= 1024 && D_UN <= 6, "ten bits per task entry in Grp::epk");
        struct Grp {
            unsigned long long epk;  // the entries of the tasks, ten bits each (an array here would be indexed by the slow-task loop
                                     // and drag the whole group into scratch memory)
            uint32_t mind[D_UN], nv, fastm, slowm;
            float s0i[D_UN];
            Raw r[D_UN];
        };
        auto grp_issue = [&](Grp &g) __attribute__((always_inline)) {
            // lane i reads the entry of task i: one LDS round trip for the whole group instead of one per field and task
            const uint32_t el = (uint32_t)(g.epk >> (10u * (lane < g.nv ? lane : 0u))) & 1023u;
            const uint4 cm = S.tmeta[el];
            const uint32_t jb = S.tblk[el];
            VCHK(lane >= g.nv || jb < ix.n_blocks, 3, jb);
            VCHK(lane >= g.nv || 8ull * cm.z < ix.blob_bytes, 4, cm.z);
#ifdef VBM25_CHECK
            if (lane < g.nv && jb < ix.n_blocks) {  // the entry is the block it names (a stale or torn entry is not)
                const uint4 gm = ix.blk_meta[jb];
                VCHK(gm.x == cm.x && gm.y == cm.y && gm.z == cm.z && gm.w == cm.w, 5, el);
            }
#endif
            const float sv = S.t_s0i[S.tterm[el]];
            g.fastm = 0;
            g.slowm = 0;
#pragma unroll
            for (int i = 0; i < D_UN; ++i) {
                const uint32_t cx = (uint32_t)__builtin_amdgcn_readlane((int)cm.x, i), cy = (uint32_t)__builtin_amdgcn_readlane((int)cm.y, i);
                const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)cm.w, i);
                const uint32_t jj = min((uint32_t)__builtin_amdgcn_readlane((int)jb, i), ix.n_blocks - 1u);
                g.mind[i] = cx;
                g.s0i[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), i));
                const bool fast = rel16_block(cx, cy, cw) && tfn_block(cw);
                g.fastm |= ((uint32_t)i < g.nv && fast ? 1u : 0u) << i;
                g.slowm |= ((uint32_t)i < g.nv && !fast ? 1u : 0u) << i;
                // unconditional (lanes beyond nv hold the first task's entry, the planes of other blocks hold zeros): a branch per
                // load makes the compiler wait for each one at the join
                g.r[i].rel = ix.post_rel16[64ull * jj + lane];
                g.r[i].tfn = ix.post_tfn[64ull * jj + lane];
            }
        };
        auto grp_finish = [&](const Grp &g, uint32_t wlo, uint32_t wspan) __attribute__((always_inline)) {
            if (g.fastm == (1u << D_UN) - 1u) {  // a full group of plane blocks (the common case): one straight-line block
#pragma unroll
                for (int i = 0; i < D_UN; ++i)
                    add_pair(g.s0i[i], g.mind[i] + (g.r[i].rel & 0xffffu), g.mind[i] + (g.r[i].rel >> 16), g.r[i].tfn & 0xffu,
                             (g.r[i].tfn >> 8) & 0xffu, g.r[i].tfn >> 16, true, true, wlo, wspan);
            } else {
#pragma unroll
                for (int i = 0; i < D_UN; ++i)
                    if ((g.fastm >> i) & 1u)
                        add_pair(g.s0i[i], g.mind[i] + (g.r[i].rel & 0xffffu), g.mind[i] + (g.r[i].rel >> 16), g.r[i].tfn & 0xffu,
                                 (g.r[i].tfn >> 8) & 0xffu, g.r[i].tfn >> 16, true, true, wlo, wspan);
            }
            uint32_t slow = g.slowm;
            while (slow) {
                const uint32_t i = (uint32_t)__ffs((int)slow) - 1u;
                slow &= slow - 1u;
                task_slow((uint32_t)(g.epk >> (10u * i)) & 1023u, wlo, wspan);
            }
#ifdef VBM25_PROFILE
            prof[13] += g.nv;
#endif
        };
        // essential tasks [0, cnt): strided over the waves, every one of them fetched
        auto run_all = [&](uint32_t cnt, uint32_t wlo, uint32_t wspan) __attribute__((always_inline)) {
            Grp g0, g1;
            auto entries = [&](Grp &g, uint32_t base) __attribute__((always_inline)) {
                g.nv = 0;
                g.epk = 0;
#pragma unroll
                for (int i = 0; i < D_UN; ++i) {
                    const uint32_t e = base + DNW * i;
                    if (e < cnt) {  // a prefix
                        g.epk |= (unsigned long long)e << (10 * i);
                        ++g.nv;
                    }
                }
            };
            uint32_t base = wave;
            if (base >= cnt) return;
#ifdef VBM25_PROFILE_SUB  // P1 taken apart (slots 10, 11, 14): entries + issue, wait for the words, adds
#define SUB_ISSUE(G) { PROF_T(s_a); grp_issue(G); PROF_T(s_b); PROF_ADD(10, s_a, s_b); }
#define SUB_FINISH(G) { PROF_T(s_a); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PROF_T(s_b); grp_finish(G, wlo, wspan); PROF_T(s_c); PROF_ADD(11, s_a, s_b); PROF_ADD(14, s_b, s_c); }
#else
#define SUB_ISSUE(G) grp_issue(G)
#define SUB_FINISH(G) grp_finish(G, wlo, wspan)
#endif
            entries(g0, base);
            SUB_ISSUE(g0);
            for (;;) {
                base += DNW * D_UN;
                const bool more1 = base < cnt;
                if (more1) {
                    entries(g1, base);
                    SUB_ISSUE(g1);
                }
                SUB_FINISH(g0);
                if (!more1) break;
                base += DNW * D_UN;
                const bool more0 = base < cnt;
                if (more0) {
                    entries(g0, base);
                    SUB_ISSUE(g0);
                }
                SUB_FINISH(g1);
                if (!more0) break;
            }
#undef SUB_ISSUE
#undef SUB_FINISH
        };
        // non-essential tasks [first, first + cnt) of one term, a contiguous share per wave: one lane per task tests
        // whether any document of the block's span can still reach the threshold (largest accumulator of the
        // 64-document buckets the span touches + rem_i, the scaled bounds of the terms not yet complete, + the
        // block's own bound); the blocks that pass are fetched, the others never touched.
        auto run_tested = [&](uint32_t first, uint32_t cnt, uint32_t theta_i, uint32_t wlo, uint32_t wspan) __attribute__((always_inline)) {
            const uint32_t share = (cnt + DNW - 1) / DNW;
          for (uint32_t o0 = 0; o0 < share; o0 += 64) {  // (a share above 64: phases of many terms)
            const uint32_t o = wave * share + o0 + lane;
            bool alive = false;
            if (o0 + lane < share && o < cnt) {
                const uint2 mm = *reinterpret_cast<const uint2 *>(&S.tmeta[first + o]);
                const uint32_t a = (max(mm.x, wlo) - wlo) >> 6, b = min(mm.y - wlo, wspan - 1u) >> 6;
                alive = true;
                if (b - a < D_SPAN_TEST / 64) {
                    uint32_t mx = 0;
                    for (uint32_t x = a; x <= b; ++x) mx = max(mx, S.bmax[x]);
                    alive = mx + (S.t_rem[S.tterm[first + o]] + S.tub[first + o]) >= theta_i;  // every addend < 2^31: no wrap
                }
            }
            unsigned long long mask = __ballot(alive);
#ifdef VBM25_PROFILE
#ifndef VBM25_PROFILE_SUB
            prof[14] += (uint32_t)__popcll(__ballot(o0 + lane < share && o < cnt)) - (uint32_t)__popcll(mask);
#endif
#endif
            while (mask) {
                Grp g;
                g.nv = 0;
                g.epk = 0;
#pragma unroll
                for (int i = 0; i < D_UN; ++i)
                    if (mask) {
                        g.epk |= (unsigned long long)(first + wave * share + o0 + (uint32_t)__ffsll((long long)mask) - 1u) << (10 * i);
                        mask &= mask - 1ull;
                        ++g.nv;
                    }
                grp_issue(g);
                grp_finish(g, wlo, wspan);
            }
          }
        };

        // ---- exact score of one candidate per lane (all 64 lanes call; `cand` marks the lanes that hold one)
        auto resolve = [&](bool cand, uint32_t d) {
            const double thd = __longlong_as_double((long long)theta_now());
            // pass 1: block upper bounds (search.rs:177-203)
            double bound = 0.0;
            for (uint32_t t = 0; t < m; ++t) {
                const uint32_t b1 = S.t_end[t];
                if (cand) {
                    const uint32_t b = r_first_block_ge(ix, S.t_b0[t], b1, d);
                    if (b < b1 && cold_args()->ix.blk_min_doc[b] <= d) bound += ix.blk_ub[b];
                }
            }
            cand = cand && bound * (1.0 + 1e-12) >= thd;
            if (!__ballot(cand)) return;
#ifdef VBM25_PROFILE
#ifndef VBM25_PROFILE_SUB
            prof[11] += (unsigned long long)__popcll(__ballot(cand));
#endif
#endif
            // pass 2: the exact score, terms in ascending key order (evaluate.rs:43-72)
            uint32_t *scr = S.scr[wave];
            double acc = 0.0;
            for (uint32_t t = 0; t < m; ++t) {
                double c = 0.0;
                const uint32_t b1 = S.t_end[t];
                uint32_t b = NONE32;
                bool pend = false;
                if (cand) {
                    b = r_first_block_ge(ix, S.t_b0[t], b1, d);
                    pend = b < b1 && cold_args()->ix.blk_min_doc[b] <= d;
                }
                for (;;) {
                    const unsigned long long pmask = __ballot(pend);
                    if (!pmask) break;
                    const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)b, __ffsll((long long)pmask) - 1);
                    VCHK(blk < ix.n_blocks, 8, blk);
                    const uint4 bm = uni4(ix.blk_meta[blk]);
                    const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
                    uint32_t a0, a1;
                    decode_doc_ids(cold_args()->ix.blob + 8ull * bm.z, md, n, bm.x, lane, a0, a1);
                    __builtin_amdgcn_wave_barrier();
                    *reinterpret_cast<uint2 *>(&scr[2 * lane]) = make_uint2(2 * lane < n ? a0 : NONE32, 2 * lane + 1 < n ? a1 : NONE32);
                    __builtin_amdgcn_wave_barrier();
                    if (pend && b == blk) {  // every lane whose document lies in this block
                        uint32_t idx = 0;
#pragma unroll
                        for (int sft = 64; sft > 0; sft >>= 1)
                            if (scr[idx + sft - 1] < d) idx += sft;
                        if (scr[idx] == d) {
                            const uint8_t *tbody = cold_args()->ix.blob + 8ull * bm.z + ((payload_bytes(md, n) + 7u) & ~7u);
                            const FieldAddr fa = field_addr(mt, n, idx);
                            const uint32_t flo = *reinterpret_cast<const uint32_t *>(tbody + fa.off0);
                            const uint32_t fhi = *reinterpret_cast<const uint32_t *>(tbody + fa.off1);
                            const uint32_t fn = cold_args()->ix.post_fn[128ull * blk + idx];
                            const double tf = (double)field_val(flo, fhi, fa);
                            c = (tf * S.t_s0[t]) / (tf + S.s1[fn]);  // Cache::evaluate, bm25.rs:355-358
                        }
                        pend = false;
                    }
                }
                acc += c;  // absent terms add 0.0 (exact)
            }
            offer(cand, acc, d);
        };
        // ---- flush of the candidate buffers (all waves, each on its own buffer): drop what the threshold has
        // overtaken, re-score the rest exactly.  No barrier inside: buffer, scratch and top-k are the wave's own.
        uint32_t cn = 0;  // entries in this wave's candidate buffer (uniform)
        auto flush = [&]() {
            PROF_T(t_fa);
            const uint32_t thi = theta_fix(theta_now());
            uint32_t w = 0;
            for (uint32_t base = 0; base < cn; base += 64) {  // in-place compaction (writes never pass the reads)
                const bool has = base + lane < cn;
                const uint32_t d = has ? S.cdoc[wave][base + lane] : 0u, v = has ? S.cval[wave][base + lane] : 0u;
                const bool keep = has && v >= thi;
                const unsigned long long km = __ballot(keep);
                const uint32_t pos = w + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                __builtin_amdgcn_wave_barrier();
                if (keep) {
                    S.cdoc[wave][pos] = d;
                    S.cval[wave][pos] = v;
                }
                w += (uint32_t)__popcll(km);
            }
            __builtin_amdgcn_wave_barrier();
            PROF_T(t_fb);
            PROF_ADD(6, t_fa, t_fb);
            VCHK(w <= (uint32_t)D_WCB && cn <= (uint32_t)D_WCB, 7, cn);
            for (uint32_t base = 0; base < w; base += 64) {
                const bool has = base + lane < w;
                resolve(has, has ? S.cdoc[wave][base + lane] : 0u);
            }
            if (lane == 0 && w) atomicAdd(&S.resolved, w);
            cn = 0;
            PROF_T(t_fc);
            PROF_ADD(7, t_fb, t_fc);
        };

        // =====================================================================
        // Window loop.  The metadata of window n + 1 is requested before P3 of window n (enum_request) and
        // consumed at the top of window n + 1: the round trip hides behind P3, the flag barrier and a flush.
        // Wave w owns the cursors of the terms w and w + 8 (registers).
        // =====================================================================
        const uint32_t wmax = uni(S.wmax);
        uint32_t W = theta_now() == 0ull ? (uint32_t)D_W0 : wmax;
        bool failed = false;
        uint32_t ocur[2] = {0, 0}, oend[2] = {0, 0}, orank[2] = {0, 0};
        bool odense[2] = {false, false};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t t = wave + DNW * s;
            if (t < m) {
                ocur[s] = uni(S.t_cur[t]);
                oend[s] = uni(S.t_end[t]);
                orank[s] = uni((uint32_t)S.t_rank[t]);
                odense[s] = uni((uint32_t)S.t_cls[t]) == 2u;
            }
        }
        uint4 em[2][2];
        double eub[2][2];
        // lane i = block cur + i of the term (chunk 1: cur + 64 + i, requested only for the dense terms: a full
        // block spans at least 128 documents, so a window holds more than 64 blocks of a term only if df > N / 2)
        auto enum_request = [&]() {
            if (wave == 0) poll_request();
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    em[s][ch] = make_uint4(NONE32, 0, 0, 0);
                    eub[s][ch] = 0.0;
                    const uint32_t jj = ocur[s] + 64u * ch + lane;
                    if (wave + DNW * s < m && jj < oend[s] && (ch == 0 || odense[s])) {
                        em[s][ch] = ix.blk_meta[jj];
                        eub[s][ch] = ix.blk_ub[jj];
                    }
                }
        };
        PROF_T(t_loop);
        PROF_ADD(8, t_item, t_loop);
        enum_request();
        // h_lay: the number of head terms the task list of this window is laid out for = the split wave 0 took in the PREVIOUS
        // window (0 in an item's first).  Any set of terms may be tested block by block -- the test stands on its own (accumulated
        // sums + the bounds of the other tested terms + the block's bound against a lower bound of the final threshold); the
        // MaxScore split only says for which terms the test pays.  The threshold only rises, so the heads only grow: the bounds
        // t_rem and the floor of the bucket maxima, which follow the NEWEST split, cover the laid-out set.
        uint32_t h_lay = 0;
        for (uint32_t wlo = lo, par = 0; wlo < hi; par ^= 1u) {
            const uint32_t whi = hi - wlo > W ? wlo + W : hi;
            const uint32_t wspan = whi - wlo;
            PROF_T(t_a);

            // ---- P0: the window's blocks = the blocks that start below its end, written to the task list straight away: every
            // term takes its place with one LDS atomic -- from the bottom of the list upwards, the head terms from the top
            // downwards -- so that nobody has to know the other terms' counts: ONE barrier.
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint32_t t = wave + DNW * s;
                if (t < m) {
                    const uint32_t ecur = ocur[s];
                    uint32_t cnt = 0, fin = 0;
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const uint32_t jj = ocur[s] + 64u * ch + lane;
                        if (ch == 1) {
                            if (cnt < 64u) {  // the first chunk was not full: nothing of the second belongs to the window
                                em[s][1] = make_uint4(NONE32, 0, 0, 0);
                            } else if (!odense[s] && jj < oend[s]) {  // not requested ahead (a term just below N / 2 at a dense spot)
                                em[s][1] = ix.blk_meta[jj];
                                eub[s][1] = ix.blk_ub[jj];
                            }
                        }
                        const bool in = jj < oend[s] && em[s][ch].x < whi;  // a prefix of the lanes
                        cnt += (uint32_t)__popcll(__ballot(in));
                        fin += (uint32_t)__popcll(__ballot(in && em[s][ch].y < whi));
                    }
                    ocur[s] += fin;  // blocks that end below the window's end are done
                    const bool head = orank[s] < h_lay;
                    uint32_t taken = 0;
                    if (lane == 0) {
                        taken = head ? atomicAdd(&S.nhi[par], cnt) : atomicAdd(&S.nlo[par], cnt);
                        if (cnt >= (uint32_t)D_SEG || taken + cnt > (uint32_t)D_TCAP) S.fail = 1;  // (128 enumerated: more may follow)
                    }
                    taken = (uint32_t)__builtin_amdgcn_readfirstlane((int)taken);
                    if (taken + cnt <= (uint32_t)D_TCAP) {  // (the two ends meeting in the middle: seen after the barrier)
                        const uint32_t at = head ? (uint32_t)D_TCAP - taken - cnt : taken;
                        uint32_t fl = lane, tb = t;
                        asm volatile("" : "+v"(fl), "+s"(tb));  // (as sl in the item setup: nothing of this block hoisted to kernel entry)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) {
                            const uint32_t i = 64u * ch + fl;
                            if (i < cnt) {
                                VCHK(at + i < (uint32_t)D_TCAP, 1, at + i);
                                S.tmeta[at + i] = em[s][ch];
                                S.tblk[at + i] = ecur + i;
                                S.tub[at + i] = __double2uint_ru(eub[s][ch] * scale) + 1u;
                                S.tterm[at + i] = (uint8_t)tb;
                            }
                        }
                    }
                }
            }
            if (wave == 0) {
                poll_consume();
                __builtin_amdgcn_wave_barrier();
                // MaxScore split (search.rs:153-169): the longest prefix of the terms in ascending upper-bound order
                // whose bounds sum below the threshold is non-essential.  Of those, the HEAD terms (df >= N / 2, the
                // lowest positions) are tested block by block in one phase after everything else; the other
                // non-essential terms are fetched with the essential ones: a phase per term would fetch 23 % of the
                // blocks of a Zipf(1) query instead of 39 % (oracle/dense_model.inc) but costs a barrier and a
                // memory round trip per term.  Lane p = position p.  The NEXT window's list is laid out for this split.
                const double thd = __longlong_as_double((long long)S.theta);
                const bool below = lane < m && S.t_cum[lane + 1u] < thd;  // a prefix of the lanes (t_cum ascends)
                uint32_t pn = (uint32_t)__popcll(__ballot(below));
                if (!cold_args()->bt.ne_on && pn < m) pn = 0;
                const bool head = lane < pn && pn < m && S.t_cls[S.t_ord[lane]] == 2;
                const unsigned long long hm = __ballot(head);
                const uint32_t h = max((uint32_t)__ffsll((long long)~hm) - 1u, h_lay);  // heads below the first other term
                if (lane < h) {
                    // bounds of the other heads (the subtraction's rounding is far below one unit; + 2 covers it and
                    // the ceiling)
                    const uint32_t t = S.t_ord[lane];
                    S.t_rem[t] = __double2uint_ru((S.t_cum[h] - S.t_ub[t]) * scale) + 2u;
                }
                if (lane == 0) {
                    S.p_ne = pn;
                    S.h_ne = h;
                    S.nlo[par ^ 1u] = 0;  // the next window's counters (last read after the previous window's barrier)
                    S.nhi[par ^ 1u] = 0;
                    S.cfl[par] = 0;       // raised in this window's P3, read after the next window's barrier
                }
            }
            lds_barrier();  // tasks, counts, split and threshold visible
            const uint32_t cnt1 = uni(S.nlo[par]), cnth = uni(S.nhi[par]);
            if (uni(S.fail) || cnt1 + cnth > (uint32_t)D_TCAP || uni(S.resolved) > D_MAX_RESOLVED) {
                failed = true;  // (too many re-scorings: the sums cannot tell masses of equal scores apart -- exhaustive kernel)
                break;
            }
            if (uni(S.p_ne) >= m) break;  // no document can reach the threshold any more (search.rs:153-169 with every term)
            const uint32_t h_new = uni(S.h_ne);
            bool flush_due = uni(S.cfl[par ^ 1u]) != 0;  // some wave's buffer was half full after the last window: everybody re-scores before P3
            const uint32_t theta_i = theta_fix(theta_now());
            {   // rem + tub of any head block <= scale x (sum of the head terms' bounds) + a few units of rounding (t_rem, tub above)
                const uint32_t heads = __double2uint_ru(S.t_cum[h_new] * scale) + 8u;
                bmax_floor = theta_i > heads ? theta_i - heads : 0u;
            }
            PROF_T(t_b);
            PROF_ADD(1, t_a, t_b);

            // ---- P1: every term but the heads = the tasks at the bottom of the list
            run_all(cnt1, wlo, wspan);
            PROF_T(t_c);
            PROF_ADD(2, t_b, t_c);
            if (whi < hi) enum_request();  // the next window's metadata: in flight during P2 and P3
            // ---- P2: the head terms (the tasks at the top of the list), every block tested
            if (cnth) {
                lds_barrier();
                run_tested((uint32_t)D_TCAP - cnth, cnth, theta_i, wlo, wspan);
            }
            PROF_T(t_d);
            PROF_ADD(3, t_c, t_d);
            lds_barrier();  // every accumulator of the window complete
            PROF_T(t_e);
            PROF_ADD(4, t_d, t_e);
#ifdef VBM25_PROFILE
            prof[0] += 1;
#endif

            // ---- P3: candidates -> the wave's buffer + the query's histogram; wipe.  Wave w owns the documents
            // [2048 w, 2048 (w + 1)) of the window = the buckets 32 w .. 32 w + 31: candidates only where the bucket
            // maximum reaches the threshold.  No barrier behind it: nobody adds to these accumulators before the next window's.
            for (;;) {
                if (flush_due) flush();  // (the one call in the loop: all waves together, or this wave alone with a full buffer)
                flush_due = true;
                PROF_T(t_f);
                constexpr uint32_t BPW = D_W / 64 / DNW;  // buckets per wave
                const double inv = (1.0 / scale) * (1.0 - 1.0 / 131072.0) * S.hscale;  // accumulator -> bucket of the score's lower bound
                const uint32_t bmv = lane < BPW ? S.bmax[wave * BPW + lane] : 0u;
                uint32_t hot = (uint32_t)__ballot(bmv >= theta_i && bmv != 0u), left = 0;
                while (hot) {
                    const uint32_t bk = (uint32_t)__ffs((int)hot) - 1u;
                    hot &= hot - 1u;
                    const uint32_t i = (wave * BPW + bk) * 64u + lane;
                    const uint32_t v = (S.acc[i >> 1] >> (16u * (i & 1u))) & 0xffffu;
                    const bool cand = v >= theta_i && v != 0u;
                    const unsigned long long cm = __ballot(cand);
                    const uint32_t c = (uint32_t)__popcll(cm);
                    if (cn + c > (uint32_t)D_WCB) {  // the bucket stays whole for the next pass
                        left |= 1u << bk;
                        continue;
                    }
                    if (cand) {
                        const uint32_t pos = cn + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                        VCHK(pos < (uint32_t)D_WCB, 6, pos);
                        VCHK(wlo + i < hi, 15, wlo + i);
                        S.cdoc[wave][pos] = wlo + i;
                        S.cval[wave][pos] = v;
                        const double hb = (double)(v > m ? v - m : 0u) * inv;
                        atomicAdd(&hrow[hb >= (double)(CUR_HB - 1) ? (uint32_t)(CUR_HB - 1) : (uint32_t)hb], 1u);
                    }
                    cn += c;
#ifdef VBM25_PROFILE
#ifndef VBM25_PROFILE_SUB
                    prof[10] += c;
#endif
#endif
                }
                PROF_T(t_g);
                PROF_ADD(5, t_f, t_g);
                if (left == 0u) {  // wipe the wave's accumulators and its bucket maxima
#pragma unroll
                    for (int x = 0; x < (int)(BPW * 32 / 256); ++x)
                        *reinterpret_cast<uint4 *>(&S.acc[wave * BPW * 32u + 256u * x + 4u * lane]) = make_uint4(0, 0, 0, 0);
                    if (lane < BPW) S.bmax[wave * BPW + lane] = 0u;
                    break;
                }
                for (uint32_t bk = 0; bk < BPW; ++bk)  // the buffer is full: the buckets taken are wiped, the wave re-scores, the rest follows
                    if (!((left >> bk) & 1u)) {
                        if (lane < 32u) S.acc[(wave * BPW + bk) * 32u + lane] = 0u;
                        if (lane == 0) S.bmax[wave * BPW + bk] = 0u;
                    }
                __builtin_amdgcn_wave_barrier();
            }
            if (cn >= (uint32_t)D_WCB / 2 && lane == 0) S.cfl[par] = 1;
            wlo = whi;
            W = min(2u * W, wmax);
            h_lay = h_new;
        }
        if (!failed) flush();
#ifdef VBM25_PROFILE
        prof[12] += 1;
        prof[9] += __builtin_readcyclecounter() - t_loop;
#endif
        if (failed) {  // hand the item to scan_many_kernel; leave LDS clean
            __syncthreads();
            // (the thread number rebuilt from the lane count: threadIdx.x itself would have to be kept -- in scratch -- for this path)
            uint32_t t0 = wave * 64u + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(t0));
            for (uint32_t i = t0; i < (uint32_t)D_W / 2; i += DWG) S.acc[i] = 0u;
            if (t0 < (uint32_t)D_W / 64) S.bmax[t0] = 0u;
        }

        // ---- item result: one list per wave
        {
            const uint32_t n = failed ? 0u : rtop.cnt;
            const KernArgsP ce = cold_args();
            const size_t list = (size_t)item * ce->bt.lpi + wave;
            VCHK(item < ce->bt.max_items && ce->bt.lpi == (uint32_t)DNW, 9, item);
            VCHK(n <= k, 14, n);
#pragma unroll
            for (int r = 0; r < RK; ++r)
                if (r * 64 + lane < n) {
                    ce->bt.res_score[list * k + r * 64 + lane] = rtop.score[r];
                    ce->bt.res_doc[list * k + r * 64 + lane] = rtop.doc[r];
                }
            if (lane == 0) {
                ce->bt.res_cnt[list] = n;
                if (wave == 0) {
                    ce->bt.item_failed[item] = failed ? 0x140u : 0u;
                    if (failed) *ce->bt.fail_any = 1u;
                }
            }
        }
    }
#ifdef VBM25_PROFILE
    if (bt.prof && lane == 0) {
        unsigned long long *o = bt.prof + ((size_t)blockIdx.x * DNW + wave) * 16;
        for (int i = 0; i < 15; ++i) o[i] = prof[i];
        o[15] = __builtin_readcyclecounter() - prof_t0;
    }
#endif
}
