// scan_dense.h -- scan_dense_kernel: queries with many postings per document (Zipf head terms; BASELINE config
// C5), <= D_T indexed terms, k <= REG_K.  Part of libvbm25's single device translation unit: included by
// search.hip inside namespace vbm25, after scan_range.h (whose helpers it uses).
//
// One 8-wave workgroup per work item (query x doc range), persistent, items from bt.work_ctr[1].  The item is
// cut into WINDOWS of <= D_W consecutive documents with one f32 accumulator per document in LDS.  A window is
//
//   P0  enumerate: the blocks of every term that start below the window's end (one lane per block: 16-byte
//       metadata + block upper bound), laid out as TASKS in descending order of the terms' token upper bounds.
//   P1  ESSENTIAL terms (search.rs:153-169: the MaxScore split on token upper bounds against the threshold):
//       every task decoded -- ids, term frequencies, fieldnorm bytes -- and an UPPER BOUND of each posting's
//       score (f32, rounded up) added to its document's accumulator with an LDS float atomic, in any order:
//       no barrier between terms, all eight waves busy, four blocks per wave in flight.
//   P2  NON-ESSENTIAL terms, one phase per term in descending order of upper bound.  A block is fetched only
//       if some document of its span can still reach the threshold: max accumulator over the span + the bounds
//       of the terms not yet complete, with the block's own upper bound for its term (search.rs:177-203: the
//       block-max test).  Every other block is SKIPPED: no id, tf or fieldnorm byte of it is read.
//   P3  candidates: documents whose accumulated bound reaches the threshold (a handful per window once the
//       threshold has settled); accumulators wiped in the same pass.
//   P4  candidates re-scored EXACTLY: per (candidate, term) the block is found among the window's tasks, decoded
//       by one wave, the posting's f64 Cache::evaluate (bm25.rs:355-358) taken; sum in ascending key order
//       (evaluate.rs:43-72) -> the register top-k of wave 0 -> the query's shared threshold.
//
// The f32 sums only SELECT; every score that is compared, kept or returned is the exact f64 sum, so results are
// bit-identical to the other kernels'.  Bounds: s0 is rounded up and carries a factor 1 + 2^-19 (five f32
// roundings + v_rcp_f32's 1 ulp < 2^-21 relative), s1 is rounded down, and every comparison of an f32 sum with
// the threshold (rounded down to f32) carries a factor 1 + 2^-17 (<= 128 order-free f32 additions).
// oracle/dense_model.inc is a scalar CPU model of exactly this scheme (tests/test_dense_model.py).
//
// The first window of a query whose threshold is still 0 is 256 documents wide and the width doubles from
// there: the number of candidates per window stays near k ln 2 while the threshold warms up.

constexpr int DNW = 8;
constexpr int DWG = DNW * 64;
constexpr int D_W = 8192;                // documents per window
constexpr int D_W0 = 256;                // first window while the threshold is 0
constexpr int D_T = 16;                  // indexed terms per query
constexpr int D_SEG = D_W / 128 + 4;     // blocks of one term that can start below a window's end (full blocks span >= 128 documents; + straddlers + the tail block)
constexpr int D_TCAP = D_T * D_SEG;
constexpr int D_CCAP = 64;               // candidates re-scored per round
constexpr int D_UN = 4;                  // tasks per wave in flight
constexpr uint32_t D_MAX_ROUNDS = 32;    // candidate rounds per window; beyond (masses of equal scores): scan_many_kernel
constexpr uint32_t D_SPAN_TEST = 512;    // widest block span the skip test reads (8 accumulators per lane)
constexpr uint32_t D_GRID = 512;         // persistent workgroups: 256 CUs x 2
constexpr uint32_t D_TARGET_ITEMS = 4096;

struct DenseLds {
    float acc[D_W];
    uint4 tmeta[D_TCAP];      // {min_doc, max_doc, off8, n | md << 8 | mt << 16 | wand_fn << 24}
    uint32_t tblk[D_TCAP];    // block index
    float tub[D_TCAP];        // block upper bound, rounded up
    uint8_t tterm[D_TCAP];
    double contrib[D_CCAP * D_T];
    uint32_t cand[D_CCAP];    // document - window start
    double s1[256];
    float s1f[256];           // rounded down
    double t_s0[D_T], t_ub[D_T], t_cum[D_T + 1];
    float t_s0f[D_T];         // rounded up x (1 + 2^-19)
    uint32_t t_cur[D_T], t_end[D_T], t_cnt[D_T], t_base[D_T], t_fin[D_T];
    uint8_t t_rank[D_T], t_ord[D_T];
    unsigned long long theta; // bits of a lower bound of the query's k-th best score
    uint32_t ncand, item, m, fail, p_ne;
    uint32_t scratch[64];
};

template <int KMAX>
__global__ void __launch_bounds__(DWG, 4) scan_dense_kernel(DevIndex ix, DevBatch bt) {
    static_assert(KMAX <= REG_K, "register top-k only");
    constexpr int RK = KMAX / 64;
    constexpr float SLACK = 1.0f + 1.0f / 131072.0f;
    __shared__ DenseLds S;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const uint32_t k = bt.k;
    const uint32_t n_items = *bt.n_items;
    for (uint32_t i = tid; i < 256; i += DWG) {
        const double v = ix.s1[i];
        S.s1[i] = v;
        S.s1f[i] = __double2float_rd(v);
    }
    for (uint32_t i = tid; i < (uint32_t)D_W; i += DWG) S.acc[i] = 0.0f;

    for (;;) {
        __syncthreads();  // previous item fully done with LDS
        if (tid == 0) {
            S.item = atomicAdd(&bt.work_ctr[1], 1u);
            S.ncand = 0;
            S.fail = 0;
        }
        __syncthreads();
        const uint32_t item = uni(S.item);
        if (item >= n_items) break;
        const Item it = bt.items[item];
        if (!(it.m & ITEM_DENSE) || (it.m & ~ITEM_DENSE) > (uint32_t)D_T) continue;  // the other kernels'
        const uint32_t q = uni(it.q), lo = uni(it.doc_lo), hi = uni(it.doc_hi);

        // ---- item setup (wave 0, lane t = term t): block ranges, first block at or after lo, bounds, order
        if (wave == 0) {
            uint32_t m = 0, term = NONE32;
            {
                const uint32_t qb = uni(bt.q_off[q]), qe = uni(bt.q_off[q + 1]);
                for (uint32_t base = qb; base < qe; base += 64) {  // compaction of the indexed terms (search.rs:59-61)
                    const uint32_t tt = base + lane < qe ? bt.term_ids[base + lane] : NONE32;
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
            double s0 = 0.0, tub = 0.0;
            if (act) {
                const uint32_t b0 = ix.term_first_block[term], b1 = ix.term_first_block[term + 1];
                s0 = ix.term_s0[term];
                const double wtf = (double)ix.term_wand_tf[term];
                tub = ((wtf * s0) / (wtf + S.s1[ix.term_wand_fn[term]])) * (1.0 + 1e-12);
                S.t_cur[lane] = r_first_block_ge(ix, b0, b1, lo);
                S.t_end[lane] = b1;
                S.t_s0[lane] = s0;
                S.t_s0f[lane] = __double2float_ru(s0) * (1.0f + 1.0f / 524288.0f);
                S.t_ub[lane] = tub;
            }
            uint32_t rank = 0;  // position in ascending order of the token upper bounds
            for (uint32_t t = 0; t < m; ++t) {
                const double ubt = readlane_f64(tub, t);
                if (act && (ubt < tub || (ubt == tub && t < lane))) ++rank;
            }
            if (act) {
                S.t_rank[lane] = (uint8_t)rank;
                S.t_ord[rank] = (uint8_t)lane;
            }
            double cum = 0.0;
            if (lane == 0) S.t_cum[0] = 0.0;
            for (uint32_t pp = 0; pp < m; ++pp) {
                const uint32_t owner = (uint32_t)__ffsll((long long)__ballot(act && rank == pp)) - 1u;
                cum += readlane_f64(tub, owner);
                if (lane == 0) S.t_cum[pp + 1] = cum;
            }
            if (lane == 0) {
                S.m = m;
                S.theta = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        const uint32_t m = uni(S.m);

        RegTopK<RK> rtop;
        rtop.init();
        unsigned long long published = 0;
        auto theta_now = [&]() -> unsigned long long {
            const unsigned long long th = S.theta;
            return ((unsigned long long)uni((uint32_t)(th >> 32)) << 32) | uni((uint32_t)th);
        };

        // ---- one task: decode the block, add the postings' upper bounds to their documents' accumulators
        struct Raw {
            uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
            uint32_t fn;
        };
        auto task_fetch = [&](const uint4 c, uint32_t j, Raw &r) {
            const uint32_t md = (c.w >> 8) & 0xff, mt = (c.w >> 16) & 0xff;
            const uint8_t *body = ix.blob + 8ull * c.z;
            const uint8_t *tbody = body + 16u * md;  // bit-packed: 16 bytes per bit of width (a multiple of 8)
            pair_fetch(body, md, lane, r.a0, r.a1, r.a2, r.a3);
            pair_fetch(tbody, mt, lane, r.b0, r.b1, r.b2, r.b3);
            r.fn = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * j)[lane];
        };
        auto add_pair = [&](uint32_t t, uint32_t d0, uint32_t d1, uint32_t f0, uint32_t f1, uint32_t fn, bool in0, bool in1,
                            uint32_t wlo, uint32_t wspan) {
            const float s0f = S.t_s0f[t];
            const uint32_t x0 = d0 - wlo, x1 = d1 - wlo;
            const float tf0 = (float)f0, tf1 = (float)f1;
            const float p0 = (tf0 * s0f) * __builtin_amdgcn_rcpf(tf0 + S.s1f[fn & 0xff]);
            const float p1 = (tf1 * s0f) * __builtin_amdgcn_rcpf(tf1 + S.s1f[fn >> 8]);
            if (in0 && x0 < wspan) atomicAdd(&S.acc[x0], p0);
            if (in1 && x1 < wspan) atomicAdd(&S.acc[x1], p1);
        };
        auto task_accumulate = [&](const uint4 c, uint32_t t, const Raw &r, uint32_t wlo, uint32_t wspan) {
            const uint32_t md = (c.w >> 8) & 0xff, mt = (c.w >> 16) & 0xff;
            uint32_t v0, v1, f0, f1;
            pair_extract(md, lane, r.a0, r.a1, r.a2, r.a3, v0, v1);
            pair_extract(mt, lane, r.b0, r.b1, r.b2, r.b3, f0, f1);
            const uint32_t own = v0 + v1;
            const uint32_t incl = wave_incl_scan_u32(own);
            const uint32_t d0 = c.x + (incl - own) + v0;
            add_pair(t, d0, d0 + v1, f0, f1, r.fn, true, true, wlo, wspan);
        };
        // byte-packed tail or raw block: generic, synchronous decode (rare: one call site)
        auto task_slow = [&](uint32_t e, uint32_t wlo, uint32_t wspan) {
            const uint4 c = uni4(S.tmeta[e]);
            const uint32_t j = uni(S.tblk[e]), t = uni((uint32_t)S.tterm[e]);
            const uint32_t n = c.w & 0xff, md = (c.w >> 8) & 0xff, mt = (c.w >> 16) & 0xff;
            const uint8_t *body = ix.blob + 8ull * c.z;
            uint32_t d0, d1, f0, f1;
            decode_doc_ids(body, md, n, c.x, lane, d0, d1);
            decode_fields(body + ((payload_bytes(md, n) + 7u) & ~7u), mt, n, lane, f0, f1);
            const uint32_t fn = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * j)[lane];
            add_pair(t, d0, d1, f0, f1, fn, 2 * lane < n, 2 * lane + 1 < n, wlo, wspan);
        };
        // tasks [first, first + cnt), strided over the waves, D_UN per wave in flight.  test: skip a block when no
        // document of its span can reach the threshold (rem_f: bounds of the other terms not yet complete).
        auto run_tasks = [&](uint32_t first, uint32_t cnt, bool test, float rem_f, float theta_f, uint32_t wlo, uint32_t wspan) {
            for (uint32_t base = wave; base < cnt; base += DNW * D_UN) {
                uint4 c[D_UN];
                uint32_t j[D_UN], t[D_UN];
                bool alive[D_UN], fast[D_UN];
                Raw r[D_UN];
#pragma unroll
                for (int i = 0; i < D_UN; ++i) {
                    const uint32_t o = base + DNW * i;
                    alive[i] = o < cnt;
                    const uint32_t e = first + (alive[i] ? o : 0u);
                    c[i] = uni4(S.tmeta[e]);
                    j[i] = uni(S.tblk[e]);
                    t[i] = uni((uint32_t)S.tterm[e]);
                    fast[i] = ((c[i].w >> 8) & 0xff) < 32u && ((c[i].w >> 16) & 0xff) < 32u;
                    if (test && alive[i]) {
                        const uint32_t a = max(c[i].x, wlo) - wlo, b = min(c[i].y - wlo, wspan - 1u);
                        if (b - a < D_SPAN_TEST) {
                            float mx = 0.0f;
                            for (uint32_t x = a + lane; x <= b; x += 64) mx = fmaxf(mx, S.acc[x]);
                            const float bound = (mx + (rem_f + S.tub[e])) * SLACK;
                            alive[i] = __ballot(bound >= theta_f) != 0ull;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < D_UN; ++i)
                    if (alive[i] && fast[i]) task_fetch(c[i], j[i], r[i]);
                uint32_t slow = 0;
#pragma unroll
                for (int i = 0; i < D_UN; ++i) {
                    if (alive[i] && fast[i]) task_accumulate(c[i], t[i], r[i], wlo, wspan);
                    if (alive[i] && !fast[i]) slow |= 1u << i;
                }
                while (slow) {
                    const uint32_t i = (uint32_t)__ffs((int)slow) - 1u;
                    slow &= slow - 1u;
                    task_slow(first + base + DNW * i, wlo, wspan);
                }
            }
        };

        // =====================================================================
        // Window loop
        // =====================================================================
        uint32_t W = theta_now() == 0ull ? (uint32_t)D_W0 : (uint32_t)D_W;
        bool failed = false;
        for (uint32_t wlo = lo; wlo < hi;) {
            const uint32_t whi = hi - wlo > W ? wlo + W : hi;
            const uint32_t wspan = whi - wlo;

            // ---- P0: enumerate.  Wave w takes the terms w and w + 8: lane i = block cur + i (two chunks of 64).
            if (tid == 0) {
                const unsigned long long g = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long th = S.theta;
                if (g > th) {
                    th = g;
                    S.theta = g;
                }
                // MaxScore split: the longest prefix of the terms in ascending upper-bound order whose bounds sum
                // below the threshold is non-essential
                const double thd = __longlong_as_double((long long)th);
                uint32_t p = 0;
                for (uint32_t pp = 1; pp <= m; ++pp)
                    if (S.t_cum[pp] < thd) p = pp;
                S.p_ne = bt.ne_on ? p : 0u;
            }
            uint4 em[2][2];
            double eub[2][2];
            uint32_t ecnt[2] = {0, 0}, ecur[2] = {0, 0};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint32_t t = wave + DNW * s;
                if (t < m) {
                    const uint32_t cur = uni(S.t_cur[t]), end = uni(S.t_end[t]);
                    ecur[s] = cur;
                    uint32_t cnt = 0, fin = 0;
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        em[s][ch] = make_uint4(NONE32, 0, 0, 0);
                        eub[s][ch] = 0.0;
                        if (ch == 1 && cnt < 64u) continue;  // the first chunk was not full
                        const uint32_t jj = cur + 64u * ch + lane;
                        if (jj < end) {
                            em[s][ch] = ix.blk_meta[jj];
                            eub[s][ch] = ix.blk_ub[jj];
                        }
                        const bool in = jj < end && em[s][ch].x < whi;  // a prefix of the lanes
                        cnt += (uint32_t)__popcll(__ballot(in));
                        fin += (uint32_t)__popcll(__ballot(in && em[s][ch].y < whi));
                    }
                    ecnt[s] = cnt;
                    if (lane == 0) {
                        S.t_cnt[t] = cnt;
                        S.t_fin[t] = fin;
                        if (cnt > (uint32_t)D_SEG) S.fail = 1;
                    }
                }
            }
            lds_barrier();  // counts of every term known
            {
                const uint32_t cu = lane < m ? S.t_cnt[lane] : 0u, ru = lane < m ? (uint32_t)S.t_rank[lane] : 0u;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint32_t t = wave + DNW * s;
                    if (t < m) {
                        const uint32_t rt = (uint32_t)__builtin_amdgcn_readlane((int)ru, (int)t);
                        // tasks in descending rank order: the terms of higher rank come first
                        const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(lane < m && ru > rt ? cu : 0u), 63);
                        if (lane == 0) S.t_base[t] = before;
                        if (!uni(S.fail)) {
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch) {
                                const uint32_t i = 64u * ch + lane;
                                if (i < ecnt[s]) {
                                    S.tmeta[before + i] = em[s][ch];
                                    S.tblk[before + i] = ecur[s] + i;
                                    S.tub[before + i] = __double2float_ru(eub[s][ch]);
                                    S.tterm[before + i] = (uint8_t)t;
                                }
                            }
                        }
                    }
                }
            }
            lds_barrier();  // tasks, split and threshold visible
            if (uni(S.fail)) {
                failed = true;
                break;
            }
            const uint32_t p_ne = uni(S.p_ne);
            if (p_ne >= m) break;  // no document can reach the threshold any more (search.rs:153-169 with every term)
            const float theta_f = __double2float_rd(__longlong_as_double((long long)theta_now()));

            // ---- P1: essential terms = ranks >= p_ne = the first tasks
            uint32_t ess_cnt;
            {
                const uint32_t cu = lane < m ? S.t_cnt[lane] : 0u, ru = lane < m ? (uint32_t)S.t_rank[lane] : 0u;
                ess_cnt = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(lane < m && ru >= p_ne ? cu : 0u), 63);
            }
            run_tasks(0, ess_cnt, false, 0.0f, theta_f, wlo, wspan);
            // ---- P2: non-essential terms, one phase per term, descending upper bound
            if (p_ne) {
                lds_barrier();
                for (uint32_t p = p_ne; p-- > 0;) {
                    const uint32_t t = uni((uint32_t)S.t_ord[p]);
                    const uint32_t cnt = uni(S.t_cnt[t]);
                    if (cnt == 0) continue;
                    // bounds of the terms below this one (they are complete only after their own phases)
                    run_tasks(uni(S.t_base[t]), cnt, true, __double2float_ru(S.t_cum[p]), theta_f, wlo, wspan);
                    if (p) lds_barrier();
                }
            }
            lds_barrier();  // every accumulator of the window complete

            // ---- P3 / P4: candidates, D_CCAP at a time
            if (tid < m) S.t_cur[tid] += S.t_fin[tid];
            for (uint32_t rounds = 0;;) {
                const float thf = __double2float_rd(__longlong_as_double((long long)theta_now()));
                for (uint32_t i = tid; i < wspan; i += DWG) {
                    const float v = S.acc[i];
                    if (v == 0.0f) continue;
                    if (v * SLACK >= thf) {
                        const uint32_t pos = atomicAdd(&S.ncand, 1u);
                        if (pos < (uint32_t)D_CCAP) {
                            S.cand[pos] = i;
                            S.acc[i] = 0.0f;
                        }  // else: stays for the next round
                    } else {
                        S.acc[i] = 0.0f;
                    }
                }
                lds_barrier();
                const uint32_t nc_all = uni(S.ncand);
                if (nc_all == 0) break;
                const uint32_t nc = min(nc_all, (uint32_t)D_CCAP);
                // pairs (candidate, term): one wave each
                for (uint32_t p = wave; p < nc * m; p += DNW) {
                    const uint32_t c = p / m, t = p - c * m;
                    const uint32_t d = wlo + uni(S.cand[c]);
                    const uint32_t tb = uni(S.t_base[t]), tc = uni(S.t_cnt[t]);
                    uint32_t e = NONE32;
                    for (uint32_t o = 0; o < tc; o += 64) {
                        bool hit = false;
                        if (o + lane < tc) {
                            const uint4 mm = S.tmeta[tb + o + lane];
                            hit = mm.x <= d && d <= mm.y;
                        }
                        const unsigned long long hm = __ballot(hit);
                        if (hm) {
                            e = tb + o + (uint32_t)__ffsll((long long)hm) - 1u;
                            break;
                        }
                    }
                    double val = 0.0;
                    if (e != NONE32) {
                        const uint4 bm = uni4(S.tmeta[e]);
                        const uint32_t j = uni(S.tblk[e]);
                        const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
                        const uint8_t *body = ix.blob + 8ull * bm.z;
                        const uint8_t *tbody = body + ((payload_bytes(md, n) + 7u) & ~7u);
                        uint32_t d0, d1, f0, f1;
                        const uint32_t fn = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * j)[lane];
                        if (md < 32u && mt < 32u) {
                            uint32_t a0, a1, a2, a3, b0, b1, b2, b3, v0, v1;
                            pair_fetch(body, md, lane, a0, a1, a2, a3);
                            pair_fetch(tbody, mt, lane, b0, b1, b2, b3);
                            pair_extract(md, lane, a0, a1, a2, a3, v0, v1);
                            pair_extract(mt, lane, b0, b1, b2, b3, f0, f1);
                            const uint32_t own = v0 + v1;
                            const uint32_t incl = wave_incl_scan_u32(own);
                            d0 = bm.x + (incl - own) + v0;
                            d1 = d0 + v1;
                        } else {
                            decode_doc_ids(body, md, n, bm.x, lane, d0, d1);
                            decode_fields(tbody, mt, n, lane, f0, f1);
                        }
                        const bool m0 = 2 * lane < n && d0 == d, m1 = 2 * lane + 1 < n && d1 == d;
                        if (m0 || m1) {
                            const double tf = (double)(m0 ? f0 : f1);
                            val = (tf * S.t_s0[t]) / (tf + S.s1[m0 ? (fn & 0xff) : (fn >> 8)]);  // Cache::evaluate, bm25.rs:355-358
                        }
                        const unsigned long long mm = __ballot(m0 || m1);
                        val = mm ? readlane_f64(val, (uint32_t)__ffsll((long long)mm) - 1u) : 0.0;
                    }
                    if (lane == 0) S.contrib[c * D_T + t] = val;
                }
                lds_barrier();
                if (wave == 0) {
                    const bool has0 = lane < nc;
                    double sc = 0.0;
                    uint32_t d = 0;
                    if (has0) {
                        d = wlo + S.cand[lane];
                        for (uint32_t t = 0; t < m; ++t) sc += S.contrib[lane * D_T + t];  // ascending key order; absent terms add 0.0
                    }
                    const unsigned long long th = theta_now();
                    const bool has = has0 && (unsigned long long)__double_as_longlong(sc) >= th &&
                                     (rtop.cnt < k || better(sc, d, rtop.kth_s, rtop.kth_d));
                    if (__ballot(has)) {
                        rtop.offer(has, sc, d, k, lane);
                        if (rtop.cnt >= k) {
                            const unsigned long long kb = (unsigned long long)__double_as_longlong(rtop.kth_s);
                            if (kb > published) {
                                if (lane == 0) {
                                    if (kb > S.theta) S.theta = kb;
                                    atomicMax(&bt.theta[q], kb);
                                }
                                published = kb;
                            }
                        }
                    }
                    if (lane == 0) S.ncand = 0;
                }
                lds_barrier();
                if (nc_all <= (uint32_t)D_CCAP) break;  // every candidate of the window taken (and its accumulator wiped)
                if (++rounds >= D_MAX_ROUNDS) {  // the approximate sums cannot tell equal scores apart: exhaustive kernel
                    failed = true;
                    break;
                }
            }
            if (failed) break;
            wlo = whi;
            W = min(2u * W, (uint32_t)D_W);
        }
        if (failed) {  // hand the item to scan_many_kernel; leave LDS clean
            __syncthreads();
            for (uint32_t i = tid; i < (uint32_t)D_W; i += DWG) S.acc[i] = 0.0f;
        }

        // ---- item result: wave 0's list
        if (wave == 0) {
            const uint32_t n = failed ? 0u : rtop.cnt;
            const size_t list = (size_t)item * bt.lpi;
#pragma unroll
            for (int r = 0; r < RK; ++r)
                if (r * 64 + lane < n) {
                    bt.res_score[list * k + r * 64 + lane] = rtop.score[r];
                    bt.res_doc[list * k + r * 64 + lane] = rtop.doc[r];
                }
            if (lane == 0) {
                bt.res_cnt[list] = n;
                bt.item_failed[item] = failed ? 0x140u : 0u;
            }
        }
    }
}
