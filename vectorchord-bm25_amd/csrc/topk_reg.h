// topk_reg.h -- RegTopK (top-k in registers), wave helpers, profiling macros.
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, block_fetch, topk_reg, scan_range, scan_dense, scan_many, merge.

// ---------------------------------------------------------------------------
// Running top-k of ONE wave held in registers: RK rows of 64 entries, sorted best first, entry e
// in row e / 64 at lane e % 64.  Insert = ballot/popcount for the position, DPP wave shift,
// rows chained through lane 63 -> lane 0.  No LDS traffic.
// ---------------------------------------------------------------------------
template <int RK>
struct RegTopK {
    double score[RK];
    uint32_t doc[RK];
    uint32_t cnt;
    double kth_s;   // k-th entry, uniform copies (valid once cnt == k)
    uint32_t kth_d;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int r = 0; r < RK; ++r) {
            score[r] = 0.0;
            doc[r] = NONE32;
        }
        cnt = 0;
        kth_s = 0.0;
        kth_d = 0;
    }
    // offer one candidate per lane (`has` marks validity); all 64 lanes call.  DEDUP: a candidate that is already in the
    // list -- same document, same score bits (a document that reaches the merge in two lists) -- is dropped.
    template <bool DEDUP = false>
    __device__ __forceinline__ void offer(bool has, double sc, uint32_t d, uint32_t k, uint32_t lane) {
        for (;;) {
            const bool alive = has && (cnt < k || better(sc, d, kth_s, kth_d));
            const unsigned long long mask = __ballot(alive);
            if (!mask) break;
            const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1;
            const double cs = readlane_f64(sc, leader);
            const uint32_t cd = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)leader);
            if (lane == leader) has = false;
            if (DEDUP) {
                unsigned long long same = 0;
#pragma unroll
                for (int r = 0; r < RK; ++r) same |= __ballot(r * 64 + lane < cnt && doc[r] == cd && score[r] == cs);
                if (same) continue;
            }
            uint32_t pos = 0;  // entries better than the candidate: a prefix of the list
#pragma unroll
            for (int r = 0; r < RK; ++r)
                pos += (uint32_t)__popcll(__ballot(r * 64 + lane < cnt && better(score[r], doc[r], cs, cd)));
            double carry_s = 0.0;
            uint32_t carry_d = NONE32;
#pragma unroll
            for (int r = 0; r < RK; ++r) {
                const double us = wave_shr1_f64(score[r]);
                const uint32_t ud = wave_shr1_u32(doc[r]);
                const double out_s = readlane_f64(score[r], 63);
                const uint32_t out_d = (uint32_t)__builtin_amdgcn_readlane((int)doc[r], 63);
                const uint32_t e = r * 64 + lane;
                if (e > pos) {
                    score[r] = lane == 0 ? carry_s : us;
                    doc[r] = lane == 0 ? carry_d : ud;
                } else if (e == pos) {
                    score[r] = cs;
                    doc[r] = cd;
                }
                carry_s = out_s;
                carry_d = out_d;
            }
            cnt = cnt < k ? cnt + 1 : k;
            if (cnt >= k) {
#pragma unroll
                for (int r = 0; r < RK; ++r)
                    if ((k - 1) / 64 == (uint32_t)r) {
                        kth_s = readlane_f64(score[r], (k - 1) & 63);
                        kth_d = (uint32_t)__builtin_amdgcn_readlane((int)doc[r], (int)((k - 1) & 63));
                    }
            }
        }
    }
};


// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// would wait for every global load in flight (the planner's metadata refills, the threshold
// poll); all hand-offs inside the tile loop go through LDS.
__device__ __forceinline__ uint32_t uni(uint32_t v) {  // value is wave-uniform: keep it in an SGPR
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint4 uni4(const uint4 v) {
    return make_uint4(uni(v.x), uni(v.y), uni(v.z), uni(v.w));
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef VBM25_PROFILE
#define PROF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define PROF_ADD(slot, a, b) prof[slot] += (b) - (a)
#define PROF_MARK(var) var = __builtin_readcyclecounter()
#else
#define PROF_T(var)
#define PROF_MARK(var)
#define PROF_ADD(slot, a, b)
#endif
