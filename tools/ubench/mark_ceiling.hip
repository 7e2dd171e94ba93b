// mark_ceiling.hip -- ceiling of a seen-filter formulation of the sparse posting scan on gfx950 (MI355X): what does it
// cost to plan the blocks of a document window, read their plane words and mark / detect second arrivals -- and nothing
// else (no candidate completion, no scoring, no top-k)?  The answer bounds every design that finds the documents of two
// posting lists with one LDS atomic per posting.
//
// Formulation measured here: WAVE-PRIVATE document windows.  A wave owns [tlo, tlo + W) of its work item (query x document
// range), has its own exact bitmap of W bits in LDS, plans the window with its 64 lanes (lane = term x candidate block),
// reads ONE coalesced word per lane and block (two ids relative to the block's first document, the post_rel16 plane of
// DESIGN.md section 1), marks both ids with one returning LDS atomic each and collects the second arrivals in a list.  No
// workgroup barrier, no shared filter; blocks that straddle a window boundary are read by both windows.
//
// The index is synthetic with C3's statistics (10 M documents, 30 k terms of 261 full blocks each, block spans of about
// 38 k documents, 1024 x 5-term queries per launch, four batches rotated): the plane is 2.0 GB, far beyond the caches.
// Variants (template switches) take the range test / the second-arrival collection out, to price them.
//
// Measurement trap found on the way: thousands of waves adding their counters to ONE cache line at kernel end cost more than
// the kernel itself (the timings doubled); every wave writes its own 64-byte result slot, the host adds them up.
// `./mark_ceiling team` runs only the team kernels (what tools/ubench/pmc.sh profiles).  Results: profiles/r4_mark_ceiling.txt.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o mark_ceiling mark_ceiling.hip      Run: ./mark_ceiling
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr uint32_t N_DOCS = 10'000'000, N_TERMS = 30'000, BPT = 261;  // blocks per term
constexpr uint32_t N_BLOCKS = N_TERMS * BPT;
constexpr uint32_t SPAN = 38300;      // documents between the first ids of consecutive blocks of a term
constexpr uint32_t STEP = 299;        // mean gap between the ids of a block (127 * 299 + 298 < SPAN)
constexpr uint32_t NQ = 1024, QT = 5, NBATCH = 4;
constexpr double ALGO_BYTES_PER_BLOCK = 467870368.0 / (1024.0 * 5 * 261);  // SURVEY 8(d) figure of C3, per block

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void gen_kernel(uint32_t *rel16, uint32_t *blk_min, uint32_t *blk_max) {
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N_BLOCKS) return;
    const uint32_t t = j / BPT, b = j - t * BPT;
    const uint32_t mn = b * SPAN + mix(t) % 2000u;
    const uint32_t i0 = 2 * lane, i1 = i0 + 1;
    const uint32_t r0 = i0 == 0 ? 0u : i0 * STEP + mix(j * 131u + i0) % STEP;
    const uint32_t r1 = i1 * STEP + mix(j * 131u + i1) % STEP;
    rel16[64ull * j + lane] = r1 << 16 | r0;
    if (lane == 0) blk_min[j] = mn;
    if (lane == 63) blk_max[j] = mn + r1;
}

struct Args {
    const uint32_t *rel16, *blk_min, *blk_max, *q_terms;  // q_terms: NQ x QT term numbers
    uint32_t nq, g;             // g items (equal document ranges) per query
    uint32_t *work_ctr;
    unsigned long long *out;    // [0] second arrivals, [1] blocks read, [2] windows, [3..] phase cycles of the team kernel
};

constexpr int WAVES = 4;        // independent waves per workgroup
constexpr int G = 8;            // blocks per group (unrolled)
constexpr int Q = 8;            // candidate blocks per term and window: 7 usable + 1 sentinel
constexpr int LIST = 256;
constexpr size_t OUT_SLOTS = 16384, OUT_BYTES = OUT_SLOTS * 64;  // one 64-byte result slot per wave

template <int WLOG2, bool RANGE, bool DUPS>
__global__ void __launch_bounds__(WAVES * 64, 2) mark_kernel(Args a) {
    constexpr uint32_t W = 1u << WLOG2, WORDS = W / 32;
    __shared__ uint32_t s_bm[WAVES][WORDS];
    __shared__ uint32_t s_list[WAVES][LIST];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *bm = s_bm[wave];
    for (uint32_t i = lane; i < WORDS; i += 64) bm[i] = 0;
    unsigned long long n_dup = 0, n_blk = 0, n_win = 0;
    const uint32_t n_items = a.nq * a.g;
    const uint32_t term_slot = lane / Q, off = lane % Q;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(a.work_ctr, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        const uint32_t q = item / a.g, part = item - q * a.g;
        const uint32_t lo = (uint32_t)((unsigned long long)N_DOCS * part / a.g);
        const uint32_t hi = (uint32_t)((unsigned long long)N_DOCS * (part + 1) / a.g);
        // cursors: per term the first block whose last document is >= lo (lane = term x candidate)
        uint32_t cur = 0, end = 0;
        if (term_slot < QT) {
            const uint32_t t = a.q_terms[q * QT + term_slot];
            uint32_t b0 = t * BPT, b1 = b0 + BPT;
            end = b1;
            while (b0 < b1) {
                const uint32_t mid = (b0 + b1) >> 1;
                if (a.blk_max[mid] < lo) b0 = mid + 1; else b1 = mid;
            }
            cur = b0;
        }
        uint32_t tlo = lo;
        // candidates of the first window
        uint32_t j = cur + off;
        bool valid = term_slot < QT && j < end;
        uint32_t mn = valid ? a.blk_min[j] : 0xffffffffu, mx = valid ? a.blk_max[j] : 0u;
        while (tlo < hi) {
            uint32_t thi = min(hi, tlo + W);
            // a term whose sentinel candidate starts inside the window has more blocks than the window takes: cut it there
            {
                uint32_t bnd = (off == Q - 1 && valid) ? mn : 0xffffffffu;
                for (int o = 32; o > 0; o >>= 1) bnd = min(bnd, (uint32_t)__shfl_xor((int)bnd, o));
                thi = min(thi, max(bnd, tlo + 1));
            }
            const uint32_t span = thi - tlo;
            const bool in_win = valid && off < Q - 1 && mn < thi;
            unsigned long long mask = __ballot(in_win);
            const unsigned long long done = __ballot(in_win && mx < thi);
            // lanes outside the window carry a dummy block (lane 63 is a sentinel candidate, never in a window): a group's
            // empty entries read it -- no branch per entry
            const uint32_t v_delta = in_win ? mn - tlo : 0x80000000u, v_blk = in_win ? j : 0u;
            // next window's candidates (their loads fly while this window is marked)
            cur += (uint32_t)__popcll((done >> (term_slot * Q)) & 0xffull);
            const uint32_t jn = cur + off;
            const bool validn = term_slot < QT && jn < end;
            const uint32_t mnn = validn ? a.blk_min[jn] : 0xffffffffu, mxn = validn ? a.blk_max[jn] : 0u;
            n_blk += (uint32_t)__popcll(mask);
            n_win += 1;

            uint32_t cnt = 0;
            uint32_t w[G], dl[G];
            auto issue = [&]() {
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int s = __ffsll((long long)(mask | 1ull << 63)) - 1;
                    mask &= mask - 1;
                    const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)v_blk, s);
                    dl[i] = (uint32_t)__builtin_amdgcn_readlane((int)v_delta, s);
                    w[i] = a.rel16[64ull * blk + lane];
                }
            };
            bool more = mask != 0;
            if (more) issue();
            while (more) {
                uint32_t cw[G], cd[G];
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    cw[i] = w[i];
                    cd[i] = dl[i];
                }
                more = mask != 0;
                if (more) issue();
                uint32_t x[2 * G], m[2 * G], o[2 * G];
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    x[2 * i] = cd[i] + (cw[i] & 0xffffu);
                    x[2 * i + 1] = cd[i] + (cw[i] >> 16);
                }
#pragma unroll
                for (int p = 0; p < 2 * G; ++p) {
                    m[p] = 1u << (x[p] & 31);
                    if (RANGE) {
                        if (x[p] >= span) m[p] = 0;
                    } else if (cd[p / 2] == 0x80000000u) {
                        m[p] = 0;
                    }
                    o[p] = atomicOr(&bm[(x[p] >> 5) & (WORDS - 1)], m[p]);
                }
                if (DUPS) {
                    unsigned long long dm[2 * G], any = 0;
#pragma unroll
                    for (int p = 0; p < 2 * G; ++p) {
                        dm[p] = __ballot((o[p] & m[p]) != 0);
                        any |= dm[p];
                    }
                    if (any) {
#pragma unroll
                        for (int p = 0; p < 2 * G; ++p) {
                            if (dm[p]) {
                                if ((o[p] & m[p]) != 0) {
                                    const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm[p] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm[p], 0u));
                                    if (pos < (uint32_t)LIST) s_list[wave][pos] = x[p];
                                }
                                cnt += (uint32_t)__popcll(dm[p]);
                            }
                        }
                    }
                } else {
                    uint32_t acc = 0;
#pragma unroll
                    for (int p = 0; p < 2 * G; ++p) acc |= o[p] & m[p];
                    cnt += acc != 0;
                }
            }
            n_dup += cnt;
            // wipe the window's bitmap
#pragma unroll
            for (uint32_t i = 0; i < WORDS / 256; ++i)
                reinterpret_cast<uint4 *>(bm)[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            tlo = thi;
            j = jn;
            valid = validn;
            mn = mnn;
            mx = mxn;
        }
    }
    if (lane == 0) {
        unsigned long long *o = a.out + 8ull * (blockIdx.x * WAVES + wave);
        o[0] = n_dup; o[1] = n_blk; o[2] = n_win;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same formulation with the plane words of the WHOLE next window in flight while the current one is marked (the
// first variant has one group of eight loads in flight per wave: at two waves per SIMD nothing hides HBM's latency).
// A window takes at most NG groups of G blocks; its words live in NG * G registers, the next window's in NG * G more.
// NOLOAD replaces the loads by arithmetic: the floor of the instruction stream itself (VALU + SALU + LDS atomics).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int NG = 4;
struct WinPlan {
    uint32_t v_delta, v_blk;      // per lane (= candidate slot): first document - window start, block number
    unsigned long long mask;      // the slots inside the window
    uint32_t span;
};

template <int WLOG2, bool NOLOAD>
__global__ void __launch_bounds__(WAVES * 64, 2) mark_deep_kernel(Args a) {
    constexpr uint32_t W = 1u << WLOG2, WORDS = W / 32;
    __shared__ uint32_t s_bm[WAVES][WORDS];
    __shared__ uint32_t s_list[WAVES][LIST];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *bm = s_bm[wave];
    for (uint32_t i = lane; i < WORDS; i += 64) bm[i] = 0;
    unsigned long long n_dup = 0, n_blk = 0, n_win = 0;
    const uint32_t n_items = a.nq * a.g;
    const uint32_t term_slot = lane / Q, off = lane % Q;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0) item = atomicAdd(a.work_ctr, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        const uint32_t q = item / a.g, part = item - q * a.g;
        const uint32_t lo = (uint32_t)((unsigned long long)N_DOCS * part / a.g);
        const uint32_t hi = (uint32_t)((unsigned long long)N_DOCS * (part + 1) / a.g);
        uint32_t cur = 0, end = 0;
        if (term_slot < QT) {
            const uint32_t t = a.q_terms[q * QT + term_slot];
            uint32_t b0 = t * BPT, b1 = b0 + BPT;
            end = b1;
            while (b0 < b1) {
                const uint32_t mid = (b0 + b1) >> 1;
                if (a.blk_max[mid] < lo) b0 = mid + 1; else b1 = mid;
            }
            cur = b0;
        }
        uint32_t tlo = lo;
        uint32_t j = cur + off;
        bool valid = term_slot < QT && j < end;
        uint32_t mn = valid ? a.blk_min[j] : 0xffffffffu, mx = valid ? a.blk_max[j] : 0u;
        // plan of the window that starts at tlo from the loaded candidates; requests the candidates of the one after it
        auto make_plan = [&]() -> WinPlan {
            uint32_t thi = min(hi, tlo + W);
            uint32_t bnd = (off == Q - 1 && valid) ? mn : 0xffffffffu;
            for (int o = 32; o > 0; o >>= 1) bnd = min(bnd, (uint32_t)__shfl_xor((int)bnd, o));
            thi = min(thi, max(bnd, tlo + 1));
            bool in_win = valid && off < Q - 1 && mn < thi;
            unsigned long long mask = __ballot(in_win);
            if (__popcll(mask) > NG * G) {  // more blocks than the registers take: the largest thi with <= NG * G of them
                uint32_t lo_v = tlo + 1, hi_v = thi;
                while (hi_v - lo_v > 1) {
                    const uint32_t mid = lo_v + ((hi_v - lo_v) >> 1);
                    if (__popcll(__ballot(valid && off < Q - 1 && mn < mid)) <= NG * G) lo_v = mid; else hi_v = mid;
                }
                thi = lo_v;
                in_win = valid && off < Q - 1 && mn < thi;
                mask = __ballot(in_win);
            }
            const unsigned long long done = __ballot(in_win && mx < thi);
            WinPlan p;
            p.v_delta = in_win ? mn - tlo : 0x80000000u;
            p.v_blk = in_win ? j : 0u;
            p.mask = mask;
            p.span = thi - tlo;
            cur += (uint32_t)__popcll((done >> (term_slot * Q)) & 0xffull);
            j = cur + off;
            valid = term_slot < QT && j < end;
            mn = valid ? a.blk_min[j] : 0xffffffffu;
            mx = valid ? a.blk_max[j] : 0u;
            tlo = thi;
            return p;
        };
        auto issue = [&](const WinPlan &p, uint32_t (&w)[NG * G]) {
            unsigned long long mask = p.mask;
#pragma unroll
            for (int i = 0; i < NG * G; ++i) {
                const int s = __ffsll((long long)(mask | 1ull << 63)) - 1;
                mask &= mask - 1;
                const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)p.v_blk, s);
                if (NOLOAD) w[i] = ((2 * lane + 1) * STEP + (blk & 255u)) << 16 | (2 * lane * STEP + (blk & 127u));
                else w[i] = a.rel16[64ull * blk + lane];
            }
        };
        uint32_t w0[NG * G], w1[NG * G];
        WinPlan p0 = make_plan(), p1 = p0;
        issue(p0, w0);
        for (;;) {
            const bool have_next = tlo < hi;
            if (have_next) {
                p1 = make_plan();
                issue(p1, w1);
            }
            n_blk += (uint32_t)__popcll(p0.mask);
            n_win += 1;
            uint32_t cnt = 0;
            unsigned long long mask = p0.mask;
            const uint32_t span = p0.span;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (mask == 0) break;
                uint32_t x[2 * G], m[2 * G], o[2 * G];
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int s = __ffsll((long long)(mask | 1ull << 63)) - 1;
                    mask &= mask - 1;
                    const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)p0.v_delta, s);
                    x[2 * i] = dl + (w0[g * G + i] & 0xffffu);
                    x[2 * i + 1] = dl + (w0[g * G + i] >> 16);
                }
#pragma unroll
                for (int p = 0; p < 2 * G; ++p) {
                    m[p] = 1u << (x[p] & 31);
                    if (x[p] >= span) m[p] = 0;
                    o[p] = atomicOr(&bm[(x[p] >> 5) & (WORDS - 1)], m[p]);
                }
                unsigned long long dm[2 * G], any = 0;
#pragma unroll
                for (int p = 0; p < 2 * G; ++p) {
                    dm[p] = __ballot((o[p] & m[p]) != 0);
                    any |= dm[p];
                }
                if (any) {
#pragma unroll
                    for (int p = 0; p < 2 * G; ++p) {
                        if (dm[p]) {
                            if ((o[p] & m[p]) != 0) {
                                const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm[p] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm[p], 0u));
                                if (pos < (uint32_t)LIST) s_list[wave][pos] = x[p];
                            }
                            cnt += (uint32_t)__popcll(dm[p]);
                        }
                    }
                }
            }
            n_dup += cnt;
#pragma unroll
            for (uint32_t i = 0; i < WORDS / 256; ++i)
                reinterpret_cast<uint4 *>(bm)[i * 64 + lane] = make_uint4(0, 0, 0, 0);
            if (!have_next) break;
            p0 = p1;
#pragma unroll
            for (int i = 0; i < NG * G; ++i) w0[i] = w1[i];
        }
    }
    if (lane == 0) {
        unsigned long long *o = a.out + 8ull * (blockIdx.x * WAVES + wave);
        o[0] = n_dup; o[1] = n_blk; o[2] = n_win;
    }
}

template <class K>
int run_kernel(const char *name, K kernel, int wlog2, Args a, const uint32_t *d_q, uint32_t *d_ctr, unsigned long long *d_out, int grid) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int reps = 20;
    float total = 0;
    unsigned long long h[3] = {0, 0, 0};
    for (int r = -2; r < reps; ++r) {
        a.q_terms = d_q + (size_t)((r + 2) % NBATCH) * NQ * QT;
        CHK(hipMemset(d_ctr, 0, 4));
        CHK(hipMemset(d_out, 0, OUT_BYTES));
        CHK(hipEventRecord(e0));
        kernel<<<grid, WAVES * 64>>>(a);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 0) total += ms;
        {
            static std::vector<unsigned long long> hv(OUT_SLOTS * 8);
            CHK(hipMemcpy(hv.data(), d_out, OUT_BYTES, hipMemcpyDeviceToHost));
            for (int c = 0; c < 3; ++c) h[c] = 0;
            for (size_t w = 0; w < OUT_SLOTS; ++w)
                for (int c = 0; c < 3; ++c) h[c] += hv[8 * w + c];
        }
    }
    const double ms = total / reps, algo = double(NQ) * QT * BPT * ALGO_BYTES_PER_BLOCK;
    printf("%-34s W=2^%d g=%u grid=%d  %.4f ms  blocks read %.3fx  windows %llu  second arrivals %llu  -> %.0f GB/s algorithmic = %.3f of 8 TB/s\n",
           name, wlog2, a.g, grid, ms, double(h[1]) / (double(NQ) * QT * BPT), h[2], h[0], algo / (ms * 1e-3) / 1e9,
           algo / (ms * 1e-3) / 8e12);
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// TEAM formulation: the waves of a workgroup share ONE exact bitmap of TEAM x 2^16 bits (window = TEAM x 65536 documents:
// the wider the window, the fewer blocks straddle its boundaries) and stay at four waves per SIMD.  Block j of a term
// belongs to wave j mod TEAM, so every wave plans only its own residue class (lane = term x candidate) -- no planner
// wave, no shared plan.  The marks are returning LDS atomics: a second arrival is seen by whichever wave comes second,
// no ordering needed.  Synchronisation is for the life cycle of the bitmap only (all marks of window n before it is
// wiped, the wipe before the marks of n + 1) and is done with two LDS counters, arrive early / wait late, not s_barrier.
// SYNC = 0 replaces the counters by s_barrier.
// ---------------------------------------------------------------------------------------------------------------------
template <int TEAM, int SYNC, bool NOLOAD, bool PROF = false>
__global__ void __launch_bounds__(TEAM * 64, 4) team_kernel(Args a) {
    constexpr uint32_t W = (uint32_t)TEAM << 16, WORDS = W / 32;
    constexpr int TQ = 8;  // candidates per term and round
    __shared__ uint32_t bm[WORDS];
    __shared__ uint32_t s_list[TEAM][LIST];
    __shared__ uint32_t s_sync[4];
    __shared__ uint32_t s_item;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t i = tid; i < WORDS; i += TEAM * 64) bm[i] = 0;
    if (tid < 4) s_sync[tid] = 0;
    unsigned long long n_dup = 0, n_blk = 0, n_win = 0;
    const uint32_t n_items = a.nq * a.g;
    const uint32_t term_slot = lane / TQ, off = lane % TQ;
    unsigned long long pr_marks = 0, pr_wa = 0, pr_wipe = 0, pr_wb = 0, pr_groups = 0;
    uint32_t epoch = 0;  // windows this workgroup has finished (the counters count TEAM arrivals per window)
    auto arrive = [&](int which) {
        if (SYNC) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(&s_sync[which], 1u);
        }
    };
    auto wait = [&](int which) {
        if (SYNC) {
            const uint32_t target = (epoch + 1u) * TEAM;
            while (__hip_atomic_load(&s_sync[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    for (;;) {
        __syncthreads();
        if (tid == 0) s_item = atomicAdd(a.work_ctr, 1u);
        __syncthreads();
        const uint32_t item = __builtin_amdgcn_readfirstlane(s_item);
        if (item >= n_items) break;
        const uint32_t q = item / a.g, part = item - q * a.g;
        const uint32_t lo = (uint32_t)((unsigned long long)N_DOCS * part / a.g);
        const uint32_t hi = (uint32_t)((unsigned long long)N_DOCS * (part + 1) / a.g);
        // this wave's cursor per term: its first block (j mod TEAM == wave) whose last document is >= lo
        uint32_t cur = 0, end = 0;
        if (term_slot < QT) {
            const uint32_t t = a.q_terms[q * QT + term_slot];
            uint32_t b0 = t * BPT, b1 = b0 + BPT;
            end = b1;
            while (b0 < b1) {
                const uint32_t mid = (b0 + b1) >> 1;
                if (a.blk_max[mid] < lo) b0 = mid + 1; else b1 = mid;
            }
            cur = b0 + ((wave + TEAM - b0 % TEAM) % TEAM);
        }
        uint32_t j = cur + TEAM * off;
        bool valid = term_slot < QT && j < end;
        uint32_t mn = valid ? a.blk_min[j] : 0xffffffffu, mx = valid ? a.blk_max[j] : 0u;
        for (uint32_t tlo = lo; tlo < hi; tlo += W) {
            const uint32_t thi = min(hi, tlo + W), span = thi - tlo;
            uint32_t cnt = 0;
            const unsigned long long c0 = PROF ? __builtin_readcyclecounter() : 0ull;
            for (;;) {  // rounds: up to TQ blocks per term each
                const bool in_win = valid && mn < thi;
                unsigned long long mask = __ballot(in_win);
                const unsigned long long done = __ballot(in_win && mx < thi);
                const uint32_t v_delta = in_win ? mn - tlo : 0x80000000u, v_blk = in_win ? j : 0u;
                const uint32_t ndone = (uint32_t)__popcll((done >> (term_slot * TQ)) & 0xffull);
                // a term whose TQ candidates all lie inside the window may have more there: another round
                const bool again = __ballot(term_slot < QT && ndone == TQ) != 0;
                cur += TEAM * ndone;
                j = cur + TEAM * off;
                valid = term_slot < QT && j < end;
                mn = valid ? a.blk_min[j] : 0xffffffffu;
                mx = valid ? a.blk_max[j] : 0u;
                n_blk += (uint32_t)__popcll(mask);
                uint32_t w[G], dl[G];
                auto issue = [&]() {
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int s = __ffsll((long long)(mask | 1ull << 63)) - 1;
                        mask &= mask - 1;
                        const uint32_t blk = (uint32_t)__builtin_amdgcn_readlane((int)v_blk, s);
                        dl[i] = (uint32_t)__builtin_amdgcn_readlane((int)v_delta, s);
                        if (NOLOAD) w[i] = ((2 * lane + 1) * STEP + (blk & 255u)) << 16 | (2 * lane * STEP + (blk & 127u));
                        else w[i] = a.rel16[64ull * blk + lane];
                    }
                };
                bool more = mask != 0;
                if (more) issue();
                while (more) {
                    pr_groups += 1;
                    uint32_t cw[G], cd[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        cw[i] = w[i];
                        cd[i] = dl[i];
                    }
                    more = mask != 0;
                    if (more) issue();
                    uint32_t x[2 * G], m[2 * G], o[2 * G];
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        x[2 * i] = cd[i] + (cw[i] & 0xffffu);
                        x[2 * i + 1] = cd[i] + (cw[i] >> 16);
                    }
#pragma unroll
                    for (int p = 0; p < 2 * G; ++p) {
                        m[p] = 1u << (x[p] & 31);
                        if (x[p] >= span) m[p] = 0;
                        o[p] = atomicOr(&bm[(x[p] >> 5) & (WORDS - 1)], m[p]);
                    }
                    unsigned long long dm[2 * G], any = 0;
#pragma unroll
                    for (int p = 0; p < 2 * G; ++p) {
                        dm[p] = __ballot((o[p] & m[p]) != 0);
                        any |= dm[p];
                    }
                    if (any) {
#pragma unroll
                        for (int p = 0; p < 2 * G; ++p) {
                            if (dm[p]) {
                                if ((o[p] & m[p]) != 0) {
                                    const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(dm[p] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dm[p], 0u));
                                    if (pos < (uint32_t)LIST) s_list[wave][pos] = x[p];
                                }
                                cnt += (uint32_t)__popcll(dm[p]);
                            }
                        }
                    }
                }
                if (!again) break;
            }
            n_dup += cnt;
            n_win += wave == 0;
            const unsigned long long c1 = PROF ? __builtin_readcyclecounter() : 0ull;
            arrive(0);
            wait(0);  // every mark of the window is in: (the product completes its second arrivals before this wait)
            const unsigned long long c2 = PROF ? __builtin_readcyclecounter() : 0ull;
            for (uint32_t i = tid; i < WORDS / 4; i += TEAM * 64) reinterpret_cast<uint4 *>(bm)[i] = make_uint4(0, 0, 0, 0);
            const unsigned long long c3 = PROF ? __builtin_readcyclecounter() : 0ull;
            arrive(1);
            wait(1);
            const unsigned long long c4 = PROF ? __builtin_readcyclecounter() : 0ull;
            pr_marks += c1 - c0;
            pr_wa += c2 - c1;
            pr_wipe += c3 - c2;
            pr_wb += c4 - c3;
            ++epoch;
        }
    }
    if (lane == 0) {  // one slot per wave (16 k waves adding to one cache line cost more than the kernel)
        unsigned long long *o = a.out + 8ull * (blockIdx.x * TEAM + wave);
        o[0] = n_dup; o[1] = n_blk; o[2] = n_win; o[3] = pr_marks; o[4] = pr_wa; o[5] = pr_wipe; o[6] = pr_wb; o[7] = pr_groups;
    }
}

template <class K>
int run_team(const char *name, K kernel, int team, Args a, const uint32_t *d_q, uint32_t *d_ctr, unsigned long long *d_out, int grid) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int reps = 20;
    float total = 0;
    unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = -2; r < reps; ++r) {
        a.q_terms = d_q + (size_t)((r + 2) % NBATCH) * NQ * QT;
        CHK(hipMemset(d_ctr, 0, 4));
        CHK(hipMemset(d_out, 0, OUT_BYTES));
        CHK(hipEventRecord(e0));
        kernel<<<grid, team * 64>>>(a);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 0) total += ms;
        {
            static std::vector<unsigned long long> hv(OUT_SLOTS * 8);
            CHK(hipMemcpy(hv.data(), d_out, OUT_BYTES, hipMemcpyDeviceToHost));
            for (int c = 0; c < 8; ++c) h[c] = 0;
            for (size_t w = 0; w < OUT_SLOTS; ++w)
                for (int c = 0; c < 8; ++c) h[c] += hv[8 * w + c];
        }
    }
    const double ms = total / reps, algo = double(NQ) * QT * BPT * ALGO_BYTES_PER_BLOCK;
    printf("%-34s team=%d g=%u grid=%d  %.4f ms  blocks read %.3fx  windows %llu  second arrivals %llu  -> %.0f GB/s algorithmic = %.3f of 8 TB/s\n",
           name, team, a.g, grid, ms, double(h[1]) / (double(NQ) * QT * BPT), h[2], h[0], algo / (ms * 1e-3) / 1e9,
           algo / (ms * 1e-3) / 8e12);
    const double nw = double(h[2]) * team;  // wave-windows
    printf("      per wave and window (cycles at 100 MHz counter x 24?): marks %.0f  wait A %.0f  wipe %.0f  wait B %.0f   groups per wave-window %.2f  blocks per wave-window %.2f\n",
           h[3] / nw, h[4] / nw, h[5] / nw, h[6] / nw, h[7] / nw, double(h[1]) / nw);
    return 0;
}

template <int WLOG2, bool RANGE, bool DUPS>
int run(const char *name, Args a, const uint32_t *d_q, uint32_t *d_ctr, unsigned long long *d_out, int grid) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int reps = 20;
    float total = 0;
    unsigned long long h[3] = {0, 0, 0};
    for (int r = -2; r < reps; ++r) {
        a.q_terms = d_q + (size_t)((r + 2) % NBATCH) * NQ * QT;
        CHK(hipMemset(d_ctr, 0, 4));
        CHK(hipMemset(d_out, 0, OUT_BYTES));
        CHK(hipEventRecord(e0));
        mark_kernel<WLOG2, RANGE, DUPS><<<grid, WAVES * 64>>>(a);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 0) total += ms;
        {
            static std::vector<unsigned long long> hv(OUT_SLOTS * 8);
            CHK(hipMemcpy(hv.data(), d_out, OUT_BYTES, hipMemcpyDeviceToHost));
            for (int c = 0; c < 3; ++c) h[c] = 0;
            for (size_t w = 0; w < OUT_SLOTS; ++w)
                for (int c = 0; c < 3; ++c) h[c] += hv[8 * w + c];
        }
    }
    const double ms = total / reps, algo = double(NQ) * QT * BPT * ALGO_BYTES_PER_BLOCK;
    printf("%-34s W=2^%d g=%u grid=%d  %.4f ms  blocks read %.3fx  windows %llu  second arrivals %llu  -> %.0f GB/s algorithmic = %.3f of 8 TB/s\n",
           name, WLOG2, a.g, grid, ms, double(h[1]) / (double(NQ) * QT * BPT), h[2], h[0], algo / (ms * 1e-3) / 1e9,
           algo / (ms * 1e-3) / 8e12);
    return 0;
}

int main(int argc, char **argv) {
    const bool only_team = argc > 1;  // `./mark_ceiling team`: the team kernels only (for rocprofv3 --pmc passes)
    uint32_t *d_rel, *d_min, *d_max, *d_q, *d_ctr;
    unsigned long long *d_out;
    CHK(hipMalloc(&d_rel, 256ull * N_BLOCKS));
    CHK(hipMalloc(&d_min, 4ull * N_BLOCKS));
    CHK(hipMalloc(&d_max, 4ull * N_BLOCKS));
    CHK(hipMalloc(&d_q, 4ull * NQ * QT * NBATCH));
    CHK(hipMalloc(&d_ctr, 4));
    CHK(hipMalloc(&d_out, OUT_BYTES));
    gen_kernel<<<(N_BLOCKS + 3) / 4, 256>>>(d_rel, d_min, d_max);
    CHK(hipDeviceSynchronize());
    std::vector<uint32_t> hq(NQ * QT * NBATCH);
    uint32_t s = 12345;
    for (size_t i = 0; i < hq.size(); i += QT) {
        for (uint32_t t = 0; t < QT;) {  // distinct terms per query
            s = s * 1664525u + 1013904223u;
            const uint32_t c = (s >> 8) % N_TERMS;
            bool dup = false;
            for (uint32_t u = 0; u < t; ++u) dup |= hq[i + u] == c;
            if (!dup) hq[i + t++] = c;
        }
    }
    CHK(hipMemcpy(d_q, hq.data(), 4 * hq.size(), hipMemcpyHostToDevice));
    Args a{};
    a.rel16 = d_rel;
    a.blk_min = d_min;
    a.blk_max = d_max;
    a.nq = NQ;
    a.work_ctr = d_ctr;
    a.out = d_out;
    if (!only_team) {
    for (uint32_t g : {4u, 8u, 16u}) {
        a.g = g;
        if (run<17, true, true>("range test + second arrivals", a, d_q, d_ctr, d_out, 512)) return 1;
    }
    a.g = 8;
    if (run<17, true, false>("range test, arrivals only counted", a, d_q, d_ctr, d_out, 512)) return 1;
    if (run<17, false, false>("no range test (wrong, priced)", a, d_q, d_ctr, d_out, 512)) return 1;
    if (run<16, true, true>("range test + second arrivals", a, d_q, d_ctr, d_out, 512)) return 1;
    if (run<16, true, true>("range test + second arrivals", a, d_q, d_ctr, d_out, 1024)) return 1;
    for (uint32_t g : {4u, 8u}) {
        a.g = g;
        if (run_kernel("deep: next window in flight", mark_deep_kernel<17, false>, 17, a, d_q, d_ctr, d_out, 512)) return 1;
        if (run_kernel("deep, no loads (instruction floor)", mark_deep_kernel<17, true>, 17, a, d_q, d_ctr, d_out, 512)) return 1;
    }
    }
    for (uint32_t g : {1u, 2u, 4u}) {
        if (only_team && g != 2u) continue;
        a.g = g;
        if (run_team("team 8 x 2^16 bits, counters", team_kernel<8, 1, false>, 8, a, d_q, d_ctr, d_out, 512)) return 1;
        if (run_team("team 8, s_barrier", team_kernel<8, 0, false>, 8, a, d_q, d_ctr, d_out, 512)) return 1;
        if (run_team("team 4 x 2^16 bits, counters", team_kernel<4, 1, false>, 4, a, d_q, d_ctr, d_out, 1024)) return 1;
        if (run_team("team 2 x 2^16 bits, counters", team_kernel<2, 1, false>, 2, a, d_q, d_ctr, d_out, 2048)) return 1;
    }
    a.g = 2;
    if (run_team("team 8, counters, no loads", team_kernel<8, 1, true>, 8, a, d_q, d_ctr, d_out, 512)) return 1;
    if (run_team("team 4, counters, no loads", team_kernel<4, 1, true>, 4, a, d_q, d_ctr, d_out, 1024)) return 1;
    if (run_team("team 4, counters, phase timers on", team_kernel<4, 1, false, true>, 4, a, d_q, d_ctr, d_out, 1024)) return 1;
    return 0;
}
