// scan_tile.h -- scan_kernel: the tile formulation.
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, scan_many, block_fetch, topk_reg, scan_tile,
// scan_cursor, merge.

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
__global__ void __launch_bounds__(CWG, KMAX > REG_K ? 4 : 6) scan_kernel(DevIndex ix, DevBatch bt) {  // (the LDS top-k of k > 256 leaves room for 4)
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
                        bt.res_score[(size_t)item * bt.lpi * k + e] = rtop.score[r];
                        bt.res_doc[(size_t)item * bt.lpi * k + e] = rtop.doc[r];
                    }
                }
                if (lane == 0) bt.res_cnt[(size_t)item * bt.lpi] = rtop.cnt;
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
                bt.res_score[(size_t)item * bt.lpi * k + i] = s_top.score[i];
                bt.res_doc[(size_t)item * bt.lpi * k + i] = s_top.doc[i];
            }
            if (tid == 0) bt.res_cnt[(size_t)item * bt.lpi] = n;
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
