// merge.h -- merge_kernel: per-chunk lists -> hits.
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, block_fetch, topk_reg, scan_range, scan_dense, scan_many, merge.

// ---------------------------------------------------------------------------
// Merge of per-chunk lists -> hits
// ---------------------------------------------------------------------------
template <int KMAX>
__global__ void __launch_bounds__(64) merge_kernel(DevIndex ix, DevBatch bt) {
    __shared__ TopK<(KMAX > REG_K ? KMAX : 1)> s_top;
    constexpr int RK = KMAX <= REG_K ? KMAX / 64 : 1;
    RegTopK<RK> rtop;
    rtop.init();
    const uint32_t q = blockIdx.x, lane = threadIdx.x, k = bt.k;
    // after scan_range_kernel<..., FUSED> (which merges in the kernel): only the queries it marked -- one of their items
    // went to scan_many_kernel
    if (bt.merge_marked && bt.n_hits[q] != NONE32) return;
    if (lane == 0) s_top.count = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t i0 = bt.q_item_base[q] * bt.lpi, i1 = bt.q_item_base[q + 1] * bt.lpi;
    // the query's threshold is a lower bound of its k-th best score: entries below it cannot be among the hits (a
    // dense query's 32 lists of 100 entries mostly are)
    const unsigned long long theta = bt.theta[q];
    for (uint32_t item = i0; item < i1; ++item) {  // every list of every item of the query
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
            if constexpr (KMAX <= REG_K) rtop.template offer<true>(has, sc, d, k, lane);  // (scan_team_kernel: a document may be in two waves' lists)
            else topk_offer<(KMAX > REG_K ? KMAX : 1)>(s_top, k, has, sc, d, lane);
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
    if constexpr (KMAX <= REG_K) {
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
}
