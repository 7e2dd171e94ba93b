// Posting scan, cursor formulation (included by search.hip): queries with at most CUR_T indexed
// terms and k <= REG_K.  ONE WAVE PER ITEM, no workgroup barriers at all.
//
// The wave keeps one cursor per term (lane t = term t), like search.rs:352-396, and always
// processes the unread block with the smallest min_doc ("L").  Processing a block = decode its
// 128 document ids (two per lane), keep them in LDS (each term keeps its last TWO blocks), and
// set two hashed bits per id.  An id whose two bits were already set may belong to a document
// that an earlier block also holds ("second arrival"): it goes to the pending list.  Term
// frequencies and fieldnorms are NOT read at all for the other postings once the threshold
// exceeds every token upper bound ("hot"): a single posting cannot reach the top-k then.
//
// Invariants (tools/cursor_model.py checks them on the CPU):
//  * blocks are processed in min_doc order, so when a block of term u is processed every earlier
//    posting of its documents sits in the LAST processed block of its term: the hashed bits only
//    have to cover those blocks (they are wiped every CUR_TCLR blocks and re-set from LDS);
//  * a pending document d is complete as soon as L > d.  It is resolved when the first of its
//    blocks is about to leave LDS (min(pending) <= max_doc of the block being overwritten): at that
//    moment L > d and every block holding d is still staged.  Resolution = binary search of d in
//    the staged blocks of all terms (lane = document x term), fetch of the matching tf / fieldnorm,
//    Cache::evaluate, sum in ascending term order, done-bit on every posting found;
//  * a block processed while the threshold was NOT hot is "cold": when it leaves LDS every posting
//    without a done bit is a single-posting document and is scored on its own.
// Results per item go to res_score / res_doc / res_cnt exactly like scan_kernel's; merge_kernel
// combines the items of a query.  An item whose pending list overflows (near-identical posting
// lists) is handed to scan_many_kernel through item_failed.

constexpr int CUR_T = 8;                 // terms per query (lanes of one resolve group <= 8)
// Tuning knobs (-D...): measured on C3 -- wipe every 16 blocks: same time, every 4: +10 %; 8 Kbit bitmaps +
// 64 pending + 5 waves per SIMD (96 VGPRs, spills): +35 %.
#ifndef CUR_BM_LOG2_V
#define CUR_BM_LOG2_V 14
#define CUR_TCLR_V 8
#define CUR_PCAP_V 96
#define CUR_MINW_V 4
#endif
constexpr int CUR_BM_LOG2 = CUR_BM_LOG2_V;  // bits per hashed bitmap (two bitmaps per wave)
constexpr int CUR_BM_WORDS = (1 << CUR_BM_LOG2) / 32;
constexpr int CUR_TCLR = CUR_TCLR_V;      // blocks between two wipes of the bitmaps
constexpr int CUR_PCAP = CUR_PCAP_V;             // pending documents per wave (5 terms: 10216 B of LDS per wave, 16 waves per CU)
constexpr int CUR_HB = 256;              // score buckets of the per-query histogram
constexpr uint32_t CUR_TARGET_ITEMS = 4096;   // = the resident waves (256 CUs x 16): one long run per wave (plan_kernel)
constexpr uint32_t CUR_MIN_CHUNK_POSTINGS = 2048;
constexpr uint32_t CUR_GRID = 256 * 24;  // persistent single-wave workgroups

__host__ __device__ constexpr uint32_t cur_lds_words(uint32_t mt) {
    return mt * 2 * 4 /*sm*/ + mt * 2 * 4 /*done*/ + 2 * CUR_BM_WORDS + mt * 256 /*stage*/ + CUR_PCAP + 64 /*rb*/ +
           mt * 2 /*sblk*/;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {  // uniform result (DPP only)
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x111, 0xf, 0xf, false));  // row_shr:1
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x112, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x114, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x118, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x142, 0xa, 0xf, false));  // row_bcast:15
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ double shfl_f64(double v, uint32_t src) {  // all lanes must be active
    const int lo = __shfl(__double2loint(v), (int)src), hi = __shfl(__double2hiint(v), (int)src);
    return __hiloint2double(hi, lo);
}

template <int KMAX>
__global__ void __launch_bounds__(64, CUR_MINW_V) scan_cursor_kernel(DevIndex ix, DevBatch bt, uint32_t mt) {
    static_assert(KMAX <= REG_K, "register top-k only");
    constexpr int RK = KMAX / 64;
    constexpr uint32_t BMM = (1u << CUR_BM_LOG2) - 1u;
    extern __shared__ uint4 cur_lds[];
    uint4 *sm = cur_lds;                                             // [mt][2] {min, max, off8, n|md|mt|..}
    uint32_t *done = reinterpret_cast<uint32_t *>(sm + mt * 2);      // [mt][2][4]
    uint32_t *bm0 = done + mt * 2 * 4;                               // hashed bitmaps
    uint32_t *bm1 = bm0 + CUR_BM_WORDS;
    uint32_t *stage = bm1 + CUR_BM_WORDS;                            // [mt][2][128] document ids
    uint32_t *pend = stage + mt * 256;                               // [CUR_PCAP]
    uint32_t *rb = pend + CUR_PCAP;                                  // [64] documents being resolved
    uint32_t *sblk = rb + 64;                                        // [mt][2] block index

    const uint32_t lane = threadIdx.x;
    const uint32_t k = bt.k;
    const uint32_t n_items = *bt.n_items;
    // s1[256] spread over the lanes: entry i in register i / 64 of lane i % 64
    double s1r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s1r[i] = ix.s1[i * 64 + lane];
    auto s1_of = [&](uint32_t fn) {  // all lanes active
        const double a = shfl_f64(s1r[0], fn & 63), b = shfl_f64(s1r[1], fn & 63);
        const double c = shfl_f64(s1r[2], fn & 63), d = shfl_f64(s1r[3], fn & 63);
        return (fn & 128) ? ((fn & 64) ? d : c) : ((fn & 64) ? b : a);
    };
#ifdef VBM25_PROFILE
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif

    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(bt.work_ctr, 1u);
        item = uni(item);
        if (item >= n_items) break;
        const Item it = bt.items[item];
        if (it.m > (uint32_t)CUR_T) continue;  // more terms or dense: the other kernels'
        PROF_T(t_item);
        const uint32_t q = uni(it.q), clo = uni(it.doc_lo), chi = uni(it.doc_hi);
        uint32_t *hrow = bt.hist + (size_t)q * CUR_HB;
        unsigned long long pg = 0;  // requested values: consumed one step after the request
        uint32_t pc[4] = {0, 0, 0, 0};
        auto poll_request = [&]() {
            pg = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 4; ++i) pc[i] = __hip_atomic_load(&hrow[4 * lane + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        poll_request();  // consumed at the end of the setup

        // ---- cursors: lane t = term t (indexed terms in ascending key order)
        uint32_t m = 0, term = NONE32;
        {
            const uint32_t qb = uni(bt.q_off[q]), qe = uni(bt.q_off[q + 1]);
            if (qe - qb <= 64) {  // one load per lane, compaction of the indexed terms through LDS
                const uint32_t tt = lane < qe - qb ? bt.term_ids[qb + lane] : NONE32;
                const bool ok = tt < ix.n_terms;  // search.rs:59-61
                const unsigned long long okm = __ballot(ok);
                if (ok) rb[__builtin_amdgcn_mbcnt_hi((uint32_t)(okm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okm, 0u))] = tt;
                __builtin_amdgcn_wave_barrier();
                m = (uint32_t)__popcll(okm);
                if (lane < m) term = rb[lane];
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
        m = uni(m);
        const bool act = lane < m;
        uint32_t nb = 0, eb = 0;
        double s0 = 0.0;
        if (act) {
            const uint32_t b0 = ix.term_first_block[term], b1 = ix.term_first_block[term + 1];
            s0 = ix.term_s0[term];
            // first block whose max_doc >= clo: guess by interpolation, gallop, then bisect
            uint32_t lo_b = b0, hi_b = b1;
            if (clo != 0 && b1 > b0) {
                uint32_t g = b0 + (uint32_t)((unsigned long long)(b1 - b0) * clo / ix.n_docs);
                if (g >= b1) g = b1 - 1;
                if (ix.blk_max_doc[g] < clo) {
                    lo_b = g + 1;
                    for (uint32_t step = 1; lo_b < hi_b; step *= 4) {
                        const uint32_t p = min(lo_b + step - 1, hi_b - 1);
                        if (ix.blk_max_doc[p] < clo) lo_b = p + 1;
                        else {
                            hi_b = p;
                            break;
                        }
                    }
                } else {
                    hi_b = g;
                    for (uint32_t step = 1; lo_b < hi_b; step *= 4) {
                        const uint32_t p = hi_b - lo_b >= step ? hi_b - step : lo_b;
                        if (ix.blk_max_doc[p] >= clo) hi_b = p;
                        else {
                            lo_b = p + 1;
                            break;
                        }
                    }
                }
                while (lo_b < hi_b) {
                    const uint32_t mid = (lo_b + hi_b) >> 1;
                    if (ix.blk_max_doc[mid] < clo) lo_b = mid + 1; else hi_b = mid;
                }
            }
            nb = lo_b;
            eb = b1;
        }
        // score -> histogram bucket: linear in [0, sum of s0) (no document scores more)
        double hscale;
        {
            double sum = 0.0;
            for (uint32_t t = 0; t < m; ++t) sum += readlane_f64(s0, t);
            hscale = (double)CUR_HB / sum;
        }
        uint4 meta1 = make_uint4(NONE32, NONE32, 0, 0), meta2 = meta1;
        double ub1 = 0.0, ub2 = 0.0;  // upper bounds of the blocks meta1 / meta2 describe
        if (act && nb < eb) {
            meta1 = ix.blk_meta[nb];
            ub1 = ix.blk_ub[nb];
        }
        if (act && nb + 1 < eb) {
            meta2 = ix.blk_meta[nb + 1];
            ub2 = ix.blk_ub[nb + 1];
        }
        uint32_t pos = (act && nb < eb && meta1.x < chi) ? meta1.x : NONE32;
        uint32_t smax0 = 0, smax1 = 0;  // lane t: max_doc of term t's staged blocks
        unsigned long long sub0 = 0, sub1 = 0;  // lane t: upper bounds (bits) of term t's staged blocks
        // metadata of the block after next, requested by the lane that advanced in the previous step
        // and merged one step later (a load that is merged at once stalls the wave for its latency)
        uint4 nm = make_uint4(NONE32, NONE32, 0, 0);
        double nu = 0.0;
        uint32_t p_sel = NONE32;

        // resolve groups: one document per group of m lanes, groups never straddle a 16-lane row
        uint32_t gt = lane & 15, gj = 0;
        while (gt >= m) {
            gt -= m;
            ++gj;
        }
        const uint32_t gpr = 16u / m;  // groups per row
        const bool gvalid = gj < gpr;
        const uint32_t gdoc = (lane >> 4) * gpr + gj, gsh = lane - gt, gmask = (1u << m) - 1u;
        const uint32_t dpb = 4 * gpr;   // documents per resolve batch

        // ---- wave state
        for (uint32_t i = lane; i < 2 * CUR_BM_WORDS / 4; i += 64) reinterpret_cast<uint4 *>(bm0)[i] = make_uint4(0, 0, 0, 0);
        if (lane < mt * 2) sm[lane] = make_uint4(NONE32, 0, 0, 0);  // empty range: min > max
        uint32_t cur = 0, val0 = 0, val1 = 0, cold0 = 0, cold1 = 0;  // bit t: term t's slots
        uint32_t steps = 0, pend_cnt = 0, pend_min = NONE32;
        bool failed = false;
        // theta: bits of a lower bound of the query's k-th best score.  Sources besides this wave's
        // own list: the k-th entries other items published (bt.theta) and the histogram of the
        // documents any item accepted (bt.hist): k documents in buckets >= b put the k-th at or
        // above the lower edge of bucket b.
        unsigned long long theta = 0, published = 0;
        auto poll_consume = [&]() {
            const unsigned long long g2 = ((unsigned long long)uni((uint32_t)(pg >> 32)) << 32) | uni((uint32_t)pg);
            if (g2 > theta) theta = g2;
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
                const double edge = ((double)b / hscale) * (1.0 - 1e-12);
                const unsigned long long eb2 = (unsigned long long)__double_as_longlong(edge);
                if (eb2 > theta) theta = eb2;
            }
        };
        bool polling = false;
        RegTopK<RK> rtop;
        rtop.init();

        auto is_fast = [](uint32_t w) {  // bit-packed with d1 deltas (a full block)
            const uint32_t md = (w >> 8) & 0xff;
            return md < 32u;
        };
        auto publish = [&]() {
            if (rtop.cnt >= k) {
                const unsigned long long kb = (unsigned long long)__double_as_longlong(rtop.kth_s);
                if (kb > theta) theta = kb;
                if (kb > published) {
                    if (lane == 0) atomicMax(&bt.theta[q], kb);
                    published = kb;
                }
            }
        };
        // offer whole documents to the list; the ones that can enter it are counted in the histogram
        auto offer = [&](bool has, double sc, uint32_t d) {
            has = has && (unsigned long long)__double_as_longlong(sc) >= theta &&
                  (rtop.cnt < k || better(sc, d, rtop.kth_s, rtop.kth_d));
            if (!__ballot(has)) return;
            if (has) {
                const double hb = sc * hscale;
                const uint32_t b = hb >= (double)(CUR_HB - 1) ? (uint32_t)(CUR_HB - 1) : (uint32_t)hb;
                atomicAdd(&hrow[b], 1u);
            }
            rtop.offer(has, sc, d, k, lane);
            publish();
        };

        // ---- pending documents below L: exact scores from the staged blocks
        auto resolve = [&](uint32_t L) {
#ifdef VBM25_PROFILE
            const unsigned long long t_r0 = __builtin_readcyclecounter();
            prof[2] += 1;
#endif
            const uint32_t n = pend_cnt;
            uint32_t w = 0, kmin = NONE32;
            for (uint32_t base = 0; base < n; base += 64) {
                const bool valid = base + lane < n;
                const uint32_t pe = valid ? pend[base + lane] : NONE32;
                const bool res = valid && pe < L, keep = valid && !res;
                const unsigned long long rmask = __ballot(res), kmask = __ballot(keep);
                __builtin_amdgcn_wave_barrier();
                if (keep) {
                    pend[w + __builtin_amdgcn_mbcnt_hi((uint32_t)(kmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kmask, 0u))] = pe;
                    kmin = min(kmin, pe);
                }
                w += (uint32_t)__popcll(kmask);
                if (!rmask) continue;
                if (res) rb[__builtin_amdgcn_mbcnt_hi((uint32_t)(rmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rmask, 0u))] = pe;
                __builtin_amdgcn_wave_barrier();
                const uint32_t nr = (uint32_t)__popcll(rmask);
#ifdef VBM25_PROFILE
                prof[3] += nr;
#endif
                for (uint32_t b0 = 0; b0 < nr; b0 += dpb) {
                    const uint32_t t = gt;
                    const bool has = gvalid && b0 + gdoc < nr;
                    const uint32_t d = has ? rb[b0 + gdoc] : NONE32;
                    uint4 a = make_uint4(NONE32, 0, 0, 0), b = a;
                    if (has) {
                        a = sm[t * 2];
                        b = sm[t * 2 + 1];
                    }
                    const bool in1 = has && d >= b.x && d <= b.y;
                    const bool inr = in1 || (has && d >= a.x && d <= a.y);
                    const uint32_t sl = t * 2 + (in1 ? 1u : 0u);
                    const uint4 sj = in1 ? b : a;
                    const uint32_t *sb = stage + sl * 128;
                    uint32_t idx = 0;
#pragma unroll
                    for (int s = 64; s > 0; s >>= 1) {
                        const uint32_t v = inr ? sb[idx + s - 1] : NONE32;
                        if (v < d) idx += s;
                    }
                    const bool found = (inr ? sb[idx] : NONE32) == d && inr;
                    // loads for the posting found: tf field, fieldnorm
                    uint32_t lo = 0, hi = 0, fn = 0;
                    const FieldAddr fa = field_addr((sj.w >> 16) & 0xff, sj.w & 0xff, idx);
                    if (found) {
                        const uint32_t nj = sj.w & 0xff, mdj = (sj.w >> 8) & 0xff;
                        const uint8_t *tbody = ix.blob + 8ull * sj.z + ((payload_bytes(mdj, nj) + 7u) & ~7u);
                        lo = *reinterpret_cast<const uint32_t *>(tbody + fa.off0);
                        hi = *reinterpret_cast<const uint32_t *>(tbody + fa.off1);
                        fn = ix.post_fn[128ull * sblk[sl] + idx];
                    }
                    const unsigned long long fmask = __ballot(found);
                    const uint32_t grp = (uint32_t)(fmask >> gsh) & gmask;
                    const uint32_t first_t = grp ? (uint32_t)__ffs((int)grp) - 1u : 0u;
                    uint32_t old = 0;
                    const uint32_t dbit = 1u << (idx & 31);
                    if (found) old = atomicOr(&done[sl * 4 + (idx >> 5)], dbit);
                    // the first term that holds the document arbitrates: resolved once, whatever the
                    // number of pending entries it has
                    const unsigned long long amask = __ballot(found && t == first_t && (old & dbit));
                    const bool already = ((uint32_t)(amask >> gsh) & gmask) != 0;
                    const double s0t = shfl_f64(s0, t);
                    const double s1v = s1_of(fn);
                    const double tf = (double)field_val(lo, hi, fa);
                    const double p = found ? (tf * s0t) / (tf + s1v) : 0.0;  // Cache::evaluate, bm25.rs:355-358
                    // sum in ascending term order in the group's first lane (absent terms add 0.0)
                    double acc = p;
                    {
                        const int plo = __double2loint(p), phi = __double2hiint(p);
#define CUR_ADD(n)                                                                                     \
    if ((uint32_t)n < m) {                                                                             \
        const int xlo = __builtin_amdgcn_update_dpp(0, plo, 0x100 + n, 0xf, 0xf, false);               \
        const int xhi = __builtin_amdgcn_update_dpp(0, phi, 0x100 + n, 0xf, 0xf, false);               \
        acc += __hiloint2double(xhi, xlo);                                                             \
    }
                        CUR_ADD(1) CUR_ADD(2) CUR_ADD(3) CUR_ADD(4) CUR_ADD(5) CUR_ADD(6) CUR_ADD(7)
#undef CUR_ADD
                    }
                    offer(has && t == 0 && grp != 0 && !already, acc, d);
                }
                __builtin_amdgcn_wave_barrier();
            }
            pend_cnt = w;
            pend_min = w ? wave_min_u32(kmin) : NONE32;
#ifdef VBM25_PROFILE
            prof[4] += __builtin_readcyclecounter() - t_r0;
#endif
        };

        // ---- a cold block leaves LDS: its postings without a done bit are whole documents
        auto cold_pass = [&](uint32_t t, uint32_t s) {  // uniform
#ifdef VBM25_PROFILE
            prof[5] += 1;
            const unsigned long long t_c0 = __builtin_readcyclecounter();
#endif
            const uint32_t sl = t * 2 + s;
            const uint4 sj = uni4(sm[sl]);
            const uint32_t blkj = uni(sblk[sl]);
            const uint32_t nj = sj.w & 0xff, mdj = (sj.w >> 8) & 0xff, mtj = (sj.w >> 16) & 0xff;
            const uint2 dd = *reinterpret_cast<const uint2 *>(stage + sl * 128 + 2 * lane);
            const uint32_t dw = done[sl * 4 + (lane >> 4)];
            const bool ok0 = dd.x != NONE32 && dd.x >= clo && dd.x < chi && !((dw >> ((2 * lane) & 31)) & 1u);
            const bool ok1 = dd.y != NONE32 && dd.y >= clo && dd.y < chi && !((dw >> ((2 * lane + 1) & 31)) & 1u);
            if (!__ballot(ok0 || ok1)) return;
            const uint8_t *tbody = ix.blob + 8ull * sj.z + ((payload_bytes(mdj, nj) + 7u) & ~7u);
            const FieldAddr f0 = field_addr(mtj, nj, 2 * lane), f1 = field_addr(mtj, nj, 2 * lane + 1);
            const uint32_t l0 = *reinterpret_cast<const uint32_t *>(tbody + f0.off0);
            const uint32_t h0 = *reinterpret_cast<const uint32_t *>(tbody + f0.off1);
            const uint32_t l1 = *reinterpret_cast<const uint32_t *>(tbody + f1.off0);
            const uint32_t h1 = *reinterpret_cast<const uint32_t *>(tbody + f1.off1);
            const uint32_t fnp = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * blkj)[lane];
            const double s0t = readlane_f64(s0, t);
            const double tf0 = (double)field_val(l0, h0, f0), tf1 = (double)field_val(l1, h1, f1);
            const double p0 = (tf0 * s0t) / (tf0 + s1_of(fnp & 0xff));
            const double p1 = (tf1 * s0t) / (tf1 + s1_of(fnp >> 8));
            offer(ok0, p0, dd.x);
            offer(ok1, p1, dd.y);
#ifdef VBM25_PROFILE
            prof[15] += __builtin_readcyclecounter() - t_c0;
#endif
        };

        auto mark = [&](uint32_t d, bool in, bool &dup) {
            if (in) {
                const uint32_t h = d & BMM, g = (__umul24(d >> CUR_BM_LOG2, 97u) + d) & BMM;
                const uint32_t hb = 1u << (h & 31), gb = 1u << (g & 31);
                const uint32_t o1 = atomicOr(&bm0[h >> 5], hb), o2 = atomicOr(&bm1[g >> 5], gb);
                dup = (o1 & hb) && (o2 & gb);
            }
        };
        auto remark = [&](uint32_t d) {
            if (d != NONE32 && d >= clo && d < chi) {
                const uint32_t h = d & BMM, g = (__umul24(d >> CUR_BM_LOG2, 97u) + d) & BMM;
                atomicOr(&bm0[h >> 5], 1u << (h & 31));
                atomicOr(&bm1[g >> 5], 1u << (g & 31));
            }
        };

        poll_consume();  // requested at the top of the setup
        published = theta;
        // ---- software pipeline: {L, sel, bm, ...} describe the block of this step; its raw words
        // (f*) were requested one step earlier
        uint32_t L = row16_min_bcast(pos);
        uint32_t sel = 0, jblk = 0;
        uint4 bm = make_uint4(NONE32, NONE32, 0, 0);
        unsigned long long bub = 0;
        uint32_t flo0 = 0, fhi0 = 0, flo1 = 0, fhi1 = 0;
        auto select = [&]() {  // after L: the cursor with the smallest position
            sel = (uint32_t)__ffsll((long long)__ballot(pos == L)) - 1u;
            bm = make_uint4(L, (uint32_t)__builtin_amdgcn_readlane((int)meta1.y, (int)sel),
                            (uint32_t)__builtin_amdgcn_readlane((int)meta1.z, (int)sel),
                            (uint32_t)__builtin_amdgcn_readlane((int)meta1.w, (int)sel));
            jblk = (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)sel);
            bub = (unsigned long long)__double_as_longlong(readlane_f64(ub1, sel));
        };
        if (L != NONE32) {
            select();
            if (is_fast(bm.w)) pair_fetch(ix.blob + 8ull * bm.z, (bm.w >> 8) & 0xff, lane, flo0, fhi0, flo1, fhi1);
        }
        PROF_T(t_loop);
        PROF_ADD(8, t_item, t_loop);
        // =====================================================================
        // Main loop: one block per step
        // =====================================================================
        while (L != NONE32) {
            PROF_T(t_a);
            const uint32_t c_sel = sel, c_jblk = jblk, c_L = L;
            const uint4 c_bm = bm;
            const unsigned long long c_ub = bub;
            const uint32_t sbit = 1u << c_sel;
            const uint32_t ns = ((cur >> c_sel) & 1u) ^ 1u;  // the slot to overwrite: the older one
            const bool fast = is_fast(c_bm.w);
            // ---- raw fields of this block
            uint32_t v0 = 0, v1 = 0;
            if (fast) pair_extract((c_bm.w >> 8) & 0xff, lane, flo0, fhi0, flo1, fhi1, v0, v1);
            // threshold poll: requested first (device-scope loads are slow and loads return in order),
            // consumed at the end of the step
            if (polling) poll_request();
            // ---- advance the cursor, pick the next block and request its words: they have the
            // whole step to arrive
            if (lane == p_sel) {
                meta2 = nm;
                ub2 = nu;
            }
            if (lane == c_sel) {
                nb += 1;
                meta1 = meta2;
                ub1 = ub2;
                const uint32_t j2 = min(nb + 1, ix.n_blocks - 1);  // past the term's end: never used
                nm = ix.blk_meta[j2];
                nu = ix.blk_ub[j2];
                pos = (nb < eb && meta1.x < chi) ? meta1.x : NONE32;
            }
            p_sel = c_sel;
            L = row16_min_bcast(pos);
            if (L != NONE32) {
                select();
                if (is_fast(bm.w)) pair_fetch(ix.blob + 8ull * bm.z, (bm.w >> 8) & 0xff, lane, flo0, fhi0, flo1, fhi1);
            }
            PROF_T(t_b);
            PROF_ADD(9, t_a, t_b);

            // ---- the block leaving LDS
            if ((ns ? val1 : val0) & sbit) {
                const uint32_t xmax = (uint32_t)__builtin_amdgcn_readlane((int)(ns ? smax1 : smax0), (int)c_sel);
                if (pend_cnt && pend_min <= xmax) resolve(c_L);
                if ((ns ? cold1 : cold0) & sbit) {
                    const unsigned long long xs = ns ? sub1 : sub0;
                    const unsigned long long xub = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(xs >> 32), (int)c_sel) << 32) |
                                                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)xs, (int)c_sel);
                    if (theta <= xub) cold_pass(c_sel, ns);
                }
            }
            PROF_T(t_c);
            PROF_ADD(10, t_b, t_c);

            // ---- document ids
            uint32_t d0, d1;
            if (fast) {
                const uint32_t own = v0 + v1;
                const uint32_t incl = wave_incl_scan_u32(own);
                d0 = c_bm.x + (incl - own) + v0;
                d1 = d0 + v1;
            } else {  // raw (width 32) or a byte-packed tail block: generic path, not prefetched
                const uint32_t n = c_bm.w & 0xff;
                decode_doc_ids(ix.blob + 8ull * c_bm.z, (c_bm.w >> 8) & 0xff, n, c_bm.x, lane, d0, d1);
                if (2 * lane >= n) d0 = NONE32;
                if (2 * lane + 1 >= n) d1 = NONE32;
            }

            // ---- stage.  A block is hot when no posting of it can reach the list on its own
            // (search.rs:203: threshold vs block upper bounds): its tf / fieldnorm bytes stay unread
            const uint32_t sl = c_sel * 2 + ns;
            *reinterpret_cast<uint2 *>(stage + sl * 128 + 2 * lane) = make_uint2(d0, d1);
            const bool hot = theta > c_ub;
            if (lane == 0) {
                sm[sl] = c_bm;
                sblk[sl] = c_jblk;
            }
            if (lane < 4) done[sl * 4 + lane] = 0;
            cur ^= sbit;
            if (ns) {
                val1 |= sbit;
                cold1 = hot ? (cold1 & ~sbit) : (cold1 | sbit);
                if (lane == c_sel) {
                    smax1 = c_bm.y;
                    sub1 = c_ub;
                }
            } else {
                val0 |= sbit;
                cold0 = hot ? (cold0 & ~sbit) : (cold0 | sbit);
                if (lane == c_sel) {
                    smax0 = c_bm.y;
                    sub0 = c_ub;
                }
            }

            // ---- mark; ids whose two bits were set already are second arrivals (or collisions)
            const bool all_in = c_bm.x >= clo && c_bm.y < chi;
            const bool in0 = d0 != NONE32 && (all_in || (d0 >= clo && d0 < chi));
            const bool in1 = d1 != NONE32 && (all_in || (d1 >= clo && d1 < chi));
            bool dup0 = false, dup1 = false;
            mark(d0, in0, dup0);
            mark(d1, in1, dup1);
            const unsigned long long m0 = __ballot(dup0), m1 = __ballot(dup1);
            PROF_T(t_d);
            PROF_ADD(11, t_c, t_d);
            if (m0 | m1) {
                const uint32_t n0 = (uint32_t)__popcll(m0), nn = n0 + (uint32_t)__popcll(m1);
                if (pend_cnt + nn > (uint32_t)CUR_PCAP) resolve(c_L);  // frees the entries below L
                if (pend_cnt + nn > (uint32_t)CUR_PCAP) {
                    failed = true;  // near-identical posting lists: scan_many_kernel's dense windows
                    break;
                }
                if (dup0) pend[pend_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] = d0;
                if (dup1) pend[pend_cnt + n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] = d1;
                pend_cnt += nn;
                pend_min = min(pend_min, wave_min_u32(min(dup0 ? d0 : NONE32, dup1 ? d1 : NONE32)));
#ifdef VBM25_PROFILE
                prof[1] += nn;
#endif
            }
            PROF_T(t_e);
            PROF_ADD(12, t_d, t_e);

            // ---- every CUR_TCLR blocks: threshold poll, wipe of the bitmaps
            ++steps;
            if (polling) poll_consume();
            polling = steps % CUR_TCLR == 0 || (!hot && (steps & 3) == 0);  // cold blocks: more often
#ifdef VBM25_PROFILE
            prof[13] += hot ? 0 : 1;
#endif
            if (steps % CUR_TCLR == 0) {
                for (uint32_t i = lane; i < 2 * CUR_BM_WORDS / 4; i += 64) reinterpret_cast<uint4 *>(bm0)[i] = make_uint4(0, 0, 0, 0);
                __builtin_amdgcn_wave_barrier();
                uint2 dd[CUR_T];
#pragma unroll
                for (int t = 0; t < CUR_T; ++t) {
                    dd[t] = make_uint2(NONE32, NONE32);
                    const uint32_t c = (cur >> t) & 1u;
                    if ((uint32_t)t < m && ((c ? val1 : val0) & (1u << t)))
                        dd[t] = *reinterpret_cast<const uint2 *>(stage + (t * 2 + c) * 128 + 2 * lane);
                }
#pragma unroll
                for (int t = 0; t < CUR_T; ++t) {
                    if ((uint32_t)t >= m) break;
                    remark(dd[t].x);
                    remark(dd[t].y);
                }
            }
            PROF_T(t_g);
            PROF_ADD(14, t_e, t_g);
        }

        // ---- end of the chunk: everything pending is complete; cold blocks still staged
        if (!failed) {
            if (pend_cnt) resolve(NONE32);
            for (uint32_t t = 0; t < m; ++t) {
                const unsigned long long x0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sub0 >> 32), (int)t) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sub0, (int)t);
                const unsigned long long x1 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(sub1 >> 32), (int)t) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)sub1, (int)t);
                if ((val0 & cold0 & (1u << t)) && theta <= x0) cold_pass(t, 0);
                if ((val1 & cold1 & (1u << t)) && theta <= x1) cold_pass(t, 1);
            }
        }
#ifdef VBM25_PROFILE
        prof[0] += steps;
        prof[6] += 1;
#endif
        const uint32_t n = failed ? 0u : rtop.cnt;
#pragma unroll
        for (int r = 0; r < RK; ++r)
            if (r * 64 + lane < n) {
                bt.res_score[(size_t)item * bt.lpi * k + r * 64 + lane] = rtop.score[r];
                bt.res_doc[(size_t)item * bt.lpi * k + r * 64 + lane] = rtop.doc[r];
            }
        if (lane == 0) {
            bt.res_cnt[(size_t)item * bt.lpi] = n;
            bt.item_failed[item] = failed ? 1u : 0u;
        }
        __builtin_amdgcn_wave_barrier();
    }
#ifdef VBM25_PROFILE
    if (bt.prof && lane == 0) {
        unsigned long long *o = bt.prof + (size_t)blockIdx.x * 16;
        for (int i = 0; i < 16; ++i) o[i] = prof[i];
        o[7] = __builtin_readcyclecounter() - prof_t0;
    }
#endif
}
