// topk_lds.h -- sorted top-k list in LDS kept by one wave (k > 256, scan_many_kernel).
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, block_fetch, topk_reg, scan_range, scan_dense, scan_many, merge.

// ---------------------------------------------------------------------------
// Sorted top-k list in LDS, maintained by ONE wave.
// Order: score descending, then doc id ascending ("better").
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool better(double sa, uint32_t da, double sb, uint32_t db) {
    return sa > sb || (sa == sb && da < db);
}

template <int KMAX>
struct TopK {
    double score[KMAX];
    uint32_t doc[KMAX];
    uint32_t count;
};

// Wave-cooperative insert of (s, d); caller guarantees it qualifies.  All 64 lanes call.
template <int KMAX>
__device__ __forceinline__ void topk_insert(TopK<KMAX> &L, uint32_t k, double s, uint32_t d,
                                            uint32_t lane) {
    const uint32_t n = L.count;
    uint32_t c = 0;
    for (uint32_t i = lane; i < n; i += 64) c += better(L.score[i], L.doc[i], s, d) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    const uint32_t pos = c;
    const uint32_t newn = n < k ? n + 1 : k;
    if (pos >= newn) return;
    // shift [pos, newn-2] up by one, from the top down
    for (int base = (int)newn - 2; base >= (int)pos; base -= 64) {
        const int i = base - (int)lane;
        double ts = 0;
        uint32_t td = 0;
        const bool act = i >= (int)pos;
        if (act) {
            ts = L.score[i];
            td = L.doc[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (act) {
            L.score[i + 1] = ts;
            L.doc[i + 1] = td;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        L.score[pos] = s;
        L.doc[pos] = d;
        L.count = newn;
    }
    __builtin_amdgcn_wave_barrier();
}

// Offer up to 64 candidates (one per lane, `has` marks validity) to the list.
template <int KMAX>
__device__ __forceinline__ void topk_offer(TopK<KMAX> &L, uint32_t k, bool has, double s,
                                           uint32_t d, uint32_t lane) {
    for (;;) {
        const uint32_t n = L.count;
        bool alive = has;
        if (alive && n >= k) alive = better(s, d, L.score[k - 1], L.doc[k - 1]);
        const unsigned long long mask = __ballot(alive);
        if (!mask) break;
        const int leader = __ffsll((long long)mask) - 1;
        const double cs = __shfl(s, leader);
        const uint32_t cd = __shfl(d, leader);
        topk_insert<KMAX>(L, k, cs, cd, lane);
        if ((int)lane == leader) has = false;
    }
}
