// merge.h -- merge_kernel: per-chunk lists -> hits.
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, block_fetch, topk_reg, scan_range, scan_dense, scan_many, merge.

// ---------------------------------------------------------------------------
// Merge of per-chunk lists -> hits
// ---------------------------------------------------------------------------
template <int KMAX>
__global__ void __launch_bounds__(64) merge_kernel(DevIndex ix, DevBatch bt) {
    __shared__ TopK<(KMAX > REG_K ? KMAX : 1)> s_top;
    __shared__ uint32_t s_map[KMAX <= REG_K ? 16 * KMAX : 1];  // (list, entry) of the entries of 16 lists
    __shared__ double s_sv_score[64];  // the entries at or above the threshold, while there are at most 64 of them (the usual case:
    __shared__ uint32_t s_sv_doc[64];  // the threshold is the best k-th score any list reached) -- ranked by counting, below
    constexpr int RK = KMAX <= REG_K ? KMAX / 64 : 1;
    RegTopK<RK> rtop;
    rtop.init();
    const uint32_t q = blockIdx.x, lane = threadIdx.x, k = bt.k;
    // after scan_range_kernel<..., FUSED> (which merges in the kernel): only the queries it marked -- one of their items
    // went to scan_many_kernel
    if (bt.merge_marked && bt.n_hits[q] != NONE32) return;
    if (lane == 0) s_top.count = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t it0 = bt.fused_g ? q * bt.fused_g : bt.q_item_base[q], it1 = bt.fused_g ? it0 + bt.fused_g : bt.q_item_base[q + 1];
    const uint32_t L0 = it0 * bt.lpi, NL = (it1 - it0) * bt.lpi;
    // the query's threshold is a lower bound of its k-th best score: entries below it cannot be among the hits (a
    // dense query's 32 lists of 100 entries mostly are)
    const unsigned long long theta = bt.theta[q];
    uint32_t nsv = 0;      // survivors buffered in s_sv_*
    bool ranked = true;    // ... and nothing has gone to the register top-k yet
    if constexpr (KMAX <= REG_K) {
        // all entries of 16 lists at a time: one round trip for the counts, one for the entries (a list after the other was
        // two dependent round trips per list -- most of this kernel's time)
        for (uint32_t lb = 0; lb < NL; lb += 16) {
            uint32_t cnt = 0;
            if (lane < 16u && lb + lane < NL) cnt = min(bt.res_cnt[L0 + lb + lane], k);
            const uint32_t incl = wave_incl_scan_u32(cnt), excl = incl - cnt;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            for (uint32_t j = 0; j < cnt; ++j) s_map[excl + j] = lane << 16 | j;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t e0 = 0; e0 < total; e0 += 64) {
                bool has = e0 + lane < total;
                double sc = 0;
                uint32_t d = 0;
                if (has) {
                    const uint32_t ds = s_map[e0 + lane];
                    const size_t at = (size_t)(L0 + lb + (ds >> 16)) * k + (ds & 0xffffu);
                    sc = bt.res_score[at];
                    d = bt.res_doc[at];
                    has = (unsigned long long)__double_as_longlong(sc) >= theta;
                }
                const unsigned long long hm = __ballot(has);
                if (ranked && nsv + (uint32_t)__popcll(hm) <= 64u) {
                    if (has) {
                        const uint32_t at = nsv + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                        s_sv_score[at] = sc;
                        s_sv_doc[at] = d;
                    }
                    nsv += (uint32_t)__popcll(hm);
                    continue;
                }
                if (ranked) {  // more than 64 survivors: the buffered ones first, then everything through the register top-k
                    ranked = false;
                    __builtin_amdgcn_wave_barrier();
                    const bool hb = lane < nsv;
                    rtop.template offer<true>(hb, hb ? s_sv_score[lane] : 0.0, hb ? s_sv_doc[lane] : 0u, k, lane);
                }
                rtop.template offer<true>(has, sc, d, k, lane);  // (a document offered twice is kept once)
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (uint32_t item = L0; item < L0 + NL; ++item) {  // every list of every item of the query
            const uint32_t cnt = uni(bt.res_cnt[item]);
            if (cnt == 0) continue;
            for (uint32_t base = 0; base < cnt; base += 64) {
                bool has = base + lane < cnt;
                double sc = 0;
                uint32_t d = 0;
                if (has) {
                    sc = bt.res_score[(size_t)item * k + base + lane];
                    d = bt.res_doc[(size_t)item * k + base + lane];
                    has = (unsigned long long)__double_as_longlong(sc) >= theta;
                }
                topk_offer<(KMAX > REG_K ? KMAX : 1)>(s_top, k, has, sc, d, lane);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    auto emit = [&](uint32_t i, double sc, uint32_t d) {
        // 24-byte record written as three 64-bit words so that padding bytes are zero
        const uint16_t *pl = ix.doc_payload + 3ull * d;
        unsigned long long *out = reinterpret_cast<unsigned long long *>(bt.hits + (size_t)q * k + i);
        out[0] = (unsigned long long)__double_as_longlong(sc);
        out[1] = (unsigned long long)d | (unsigned long long)pl[0] << 32 | (unsigned long long)pl[1] << 48;
        out[2] = (unsigned long long)pl[2];
    };
    uint32_t n;
    if (KMAX <= REG_K && ranked) {
        // at most 64 survivors, one per lane: an entry's place = the number of entries better than it (score descending, ties by
        // ascending document); an entry that repeats an earlier one (same document, same score bits) is dropped and counts for nobody
        __builtin_amdgcn_wave_barrier();
        const bool mine = lane < nsv;
        const double sc = mine ? s_sv_score[lane] : 0.0;
        const uint32_t d = mine ? s_sv_doc[lane] : 0u;
        bool rep = false;
        for (uint32_t j = 0; j + 1 < nsv; ++j) {
            const double sj = readlane_f64(sc, j);
            const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)j);
            rep = rep || (lane > j && dj == d && sj == sc);
        }
        const unsigned long long live = __ballot(mine && !rep);
        uint32_t place = 0;
        for (uint32_t j = 0; j < nsv; ++j) {
            if (!((live >> j) & 1ull)) continue;
            const double sj = readlane_f64(sc, j);
            const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)j);
            place += better(sj, dj, sc, d) ? 1u : 0u;
        }
        n = min((uint32_t)__popcll(live), k);
        if (mine && !rep && place < k) emit(place, sc, d);
    } else if constexpr (KMAX <= REG_K) {
        n = rtop.cnt;
#pragma unroll
        for (int r = 0; r < RK; ++r)
            if (r * 64 + lane < n) emit(r * 64 + lane, rtop.score[r], rtop.doc[r]);
    } else {
        n = s_top.count;
        for (uint32_t i = lane; i < n; i += 64) emit(i, s_top.score[i], s_top.doc[i]);
    }
    if (lane == 0) {
        bt.n_hits[q] = n;
        if (bt.merge_marked) bt.theta[q] = 0;  // the one-launch route keeps the per-launch state clean
    }
    if (bt.merge_clean) {
        // the route without plan_kernel: this query's share of the per-launch state back to zero (the next launch starts on
        // it); what the test aids want to see afterwards is kept aside
        uint32_t nf = 0;
        for (uint32_t i = it0 + lane; i < it1; i += 64) {
            nf += bt.item_failed[i] != 0u ? 1u : 0u;
            bt.item_failed[i] = 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nf += __shfl_xor(nf, o);
        for (uint32_t i = lane; i < NL; i += 64) bt.res_cnt[L0 + i] = 0;
        if (bt.hist)
#pragma unroll
            for (int i = 0; i < CUR_HB / 64; ++i) bt.hist[(size_t)q * CUR_HB + 64 * i + lane] = 0;
        if (lane == 0) {
            bt.q_failed[q] = nf;
            bt.theta_last[q] = theta;
            bt.theta[q] = 0;
            if (q == 0) {
                bt.work_ctr[0] = 0;
                bt.work_ctr[1] = 0;
                *bt.fail_any = 0;
            }
        }
    }
}
