// scan_win.h -- scan_win_kernel: the document-WINDOW formulation of the sparse posting scan (queries of <= WN_T indexed terms
// whose lists are of comparable length; k <= 256: one, two or four register rows of the top-k): the dominant kernel of C3.
// Part of libvbm25's device code: included inside namespace vbm25 after device_types, decode, topk_lds, block_fetch, topk_reg.
//
// Replaces the traversal of search.rs:149-280 (the WAND main loop) for these queries.  What the loop computes -- the k best sums of
// Cache::evaluate over the query's terms (bm25.rs:355-358), ties by ascending document -- is reached differently:
//
//   * The document space is cut into WINDOWS of 2^16 documents.  post_id16 holds the low 16 bits of every posting's document id in
//     posting order and win_off[t][w] the number of term t's postings below document w << 16 (both derived at index creation,
//     plan.h): the postings of term t in window w are ONE contiguous run of 16-bit values whose value IS the posting's bit in the
//     window's filter.  No block metadata is read, no id is rebuilt, no tile is planned: a (term, window) pair is one coalesced
//     8-byte load per lane (four postings) and four LDS atomics.
//   * A WAVE owns its work item (a run of windows of one query) and everything it touches: an exact filter of 2^16 bits, the staged
//     ids of the current window, a list of second arrivals, its top-k in registers.  There is no workgroup barrier anywhere; a
//     workgroup is up to twelve independent waves (wn_waves).  The runs of the next two windows are in flight while this one is marked.
//   * A mark that finds its bit set is a SECOND ARRIVAL: the document has postings in two of the query's lists.  After the window's
//     marks every (second arrival, term) pair is one lane: bisection of the term's staged run, the posting's tf / fieldnorm from
//     ONE word of post_tfn, Cache::evaluate; the entry of the LAST term that holds the document sums the row in ascending key
//     order (evaluate.rs:43-72) and offers it -- entries of earlier terms see a later one and drop out, so a document is offered once.
//   * Documents with a single posting matter only where a block's upper bound (search.rs:377-380; blk_ub, evaluated once per
//     index) reaches the threshold.  That is decided at the item's END, against the threshold the windows have raised by then (the
//     COLD pass): the item's blocks whose bound still reaches it are decoded, their postings scored on their own, and a posting
//     that would enter the list is offered unless its document has a posting in another list of the query (the completion's).
//   * The threshold: theta0 from term_kth_ub (the item starts with a valid bound), the wave's own k-th score, the query's shared
//     64-bit atomicMax word polled once per window.  Filtering on score < threshold is exact (a lower bound of the final k-th
//     score; ties are kept).
//
// The window loop's loads are HAND-ISSUED (inline-asm global_load_* + s_waitcnt vmcnt(n), the wn_load_* / wn_wait_* helpers below) in
// a fixed order per window -- the shared threshold, the tf / fieldnorm words of the two open completion passes, the runs of the
// next-but-one window -- so that every wait names exactly the loads it needs.  Left to the compiler the loop waited for everything
// at its header and after every branch that held a load.  Two rules keep this safe; tools/check_inflight.py (a CPU test) checks
// them on the ISA: (1) the loads are unconditional -- the kernel is a template over the run loads per window MT, dummies below
// the query's term count -- and no compiler-generated instruction touches a register while a hand-issued load into it is in
// flight; (2) an asm load that takes its address from an SGPR pair starts with s_nop 4: the compiler does not pad the hazard between
// a VALU write of an SGPR (v_readlane, an SGPR restored from its spill lane) and a VMEM read of it inside an asm statement.
//
// Results: one list per item at res_*[item] (bt.lpi == 1 on this route); merge_kernel merges a query's lists.  An item with a
// window that collects more than WN_LIST second arrivals or a term frequency above 255 in a second arrival is handed to
// scan_many_kernel (item_failed).  A run thicker than one load per lane (WN_SLOT postings) takes a chunk loop on the spot.

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // (the wipe names M0 as clobbered: a reserved register)
constexpr int WN_T = 8;               // indexed terms per query
// Independent waves per workgroup: ONE workgroup per CU holds all the LDS its waves need (three workgroups of four waves, 3 x 54 KB,
// were never resident together: the third one ran after the others).  As many waves as the LDS holds -- a wave's share grows with
// the run loads MT it is compiled for -- and the registers allow: up to 12 waves have 168 VGPRs each, 13 to 16 have 128.
// (round 6: the instantiations that fit 128 VGPRs -- at most five run loads with one register row of the top-k, at most four with two --
// COULD run four waves on two of the CU's SIMDs, 14 per workgroup; measured with -DVBM25_WIN_WAVES_MAX=13 / 14: slower)
#ifndef VBM25_WIN_WAVES_MAX
#define VBM25_WIN_WAVES_MAX 12
#endif
constexpr int wn_wave_lds(int mt) { return 8192 + 512 * mt + 256; }  // a wave's filter, its staged runs, its list of second arrivals
constexpr int wn_waves(int mt, int rk) {
    const int by_lds = (163840 - 2048) / wn_wave_lds(mt);
    const int by_regs = (rk == 1 && mt <= 5) || (rk == 2 && mt <= 4) ? 16 : 12;  // (<= 128 VGPRs: measured, `make resources`)
    const int n = by_lds < by_regs ? by_lds : by_regs;
    return n < VBM25_WIN_WAVES_MAX ? n : VBM25_WIN_WAVES_MAX;
}
constexpr int WN_BM_WORDS = 2048;     // 2^16 bits
constexpr int WN_SLOT = 256;          // staged ids per term: what one 8-byte load per lane covers
constexpr int WN_LIST = 64;           // second arrivals per window
constexpr uint32_t WN_GRID = 256;     // persistent workgroups: one per CU

// A wave's filter lives in its own array, 8 KB-aligned: the address of a posting's word is `base | offset` -- one v_and_or_b32 --
// where a sum would take an instruction more per posting (BM below).
typedef __attribute__((address_space(3))) uint32_t wn_lds_u32;
// (the mask 0x1ffc travels in a VGPR: v_and_or_b32 takes no literal on gfx9, and one scalar operand -- the base -- at most)
struct WnBm {
    uint32_t base, mask;
};
__device__ __forceinline__ uint32_t wn_or_rtn(const WnBm bm, const uint32_t x, const uint32_t bit) {
    return __hip_atomic_fetch_or((wn_lds_u32 *)(((x >> 3) & bm.mask) | bm.base), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wn_or(const WnBm bm, const uint32_t x, const uint32_t bit) {
    (void)__hip_atomic_fetch_or((wn_lds_u32 *)(((x >> 3) & bm.mask) | bm.base), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int MT>
struct WinWave {
    alignas(16) uint16_t stage[MT * WN_SLOT];
    uint32_t list[WN_LIST];           // the window's second arrivals: a slot of the staged runs, or (bit 31) an id
};

__device__ __forceinline__ uint32_t wn_mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// The lane's four postings of a (term, window) pair: ids in v (two per word), posting r0 + j of the run's n is the lane's j-th.
// Marks them in the window's filter; RET: the ones whose bit was already there go to the list of second arrivals.
template <bool RET, int MT>
__device__ __forceinline__ void wn_pair(WinWave<MT> &S, const WnBm bmbase, const uint2 v, const uint32_t r0, const uint32_t n,
                                        const uint32_t t, uint32_t &nd) {
    const uint32_t x0 = v.x & 0xffffu, x1 = v.x >> 16, x2 = v.y & 0xffffu, x3 = v.y >> 16;
    const uint32_t b0 = r0 < n ? 1u << (x0 & 31u) : 0u;
    const uint32_t b1 = r0 + 1u < n ? 1u << (x1 & 31u) : 0u;
    const uint32_t b2 = r0 + 2u < n ? 1u << (x2 & 31u) : 0u;
    const uint32_t b3 = r0 + 3u < n ? 1u << (x3 & 31u) : 0u;
    if (!RET) {  // the window's first term: the filter is empty, nothing can be there yet
        wn_or(bmbase, x0, b0);
        wn_or(bmbase, x1, b1);
        wn_or(bmbase, x2, b2);
        wn_or(bmbase, x3, b3);
        return;
    }
    const uint32_t h0 = wn_or_rtn(bmbase, x0, b0) & b0;
    const uint32_t h1 = wn_or_rtn(bmbase, x1, b1) & b1;
    const uint32_t h2 = wn_or_rtn(bmbase, x2, b2) & b2;
    const uint32_t h3 = wn_or_rtn(bmbase, x3, b3) & b3;
    if (__ballot((h0 | h1 | h2 | h3) != 0u)) {
        uint32_t hm = (h0 ? 1u : 0u) | (h1 ? 2u : 0u) | (h2 ? 4u : 0u) | (h3 ? 8u : 0u);
        do {
            const bool has = hm != 0u;
            const uint32_t j = (uint32_t)__ffs((int)hm) - 1u;
            hm &= hm - 1u;
            const uint32_t x = j == 0u ? x0 : j == 1u ? x1 : j == 2u ? x2 : x3;
            const unsigned long long mk = __ballot(has);
            const uint32_t pos = nd + wn_mbcnt(mk);
            if (has && pos < (uint32_t)WN_LIST) S.list[pos] = x | t << 16 | 0x80000000u;  // (bit 31: the id itself, not a slot of the staged run)
            nd += (uint32_t)__popcll(mk);
        } while (__ballot(hm != 0u));
    }
}

// The loads that stay in flight across the window loop's iterations -- the runs of the window after next, the tf / fieldnorm word of
// the last window's second arrivals, the query's shared threshold -- are issued and waited for BY HAND.  Left to the compiler
// every one of them was waited for too early: its count of the loads in flight is merged conservatively where paths join (the
// loop header, every rare branch with a load inside), and `s_waitcnt vmcnt(n)` with too small an n waits for the newest loads
// too.  An asm load is invisible to that count, which only makes the compiler's own waits stricter (vmcnt counts every load);
// the waits below name the registers as in/out operands, so nothing that reads them can be scheduled above the wait.
// Issue order at the end of every window w: P(w) the threshold, G(w) the word, R(w + 2) the eight runs.
// (s_nop 4: a vector memory instruction must not read an SGPR within five wait states of a VALU instruction writing it -- v_readlane
// does, and so does the restore of a spilled SGPR.  The compiler pads its own loads; it cannot see into an asm statement: without
// the padding the kernel of eight run loads, which spills the pointer of the threshold word, faulted on a stale address.)
__device__ __forceinline__ void wn_load_run(unsigned long long &dst, const uint32_t voff, const unsigned long long sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void wn_load_word(uint32_t &dst, const uint32_t *addr) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(addr));
}
__device__ __forceinline__ void wn_load_theta(unsigned long long &dst, const uint32_t vzero, const unsigned long long *sbase) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 sc1" : "=v"(dst) : "v"(vzero), "s"(sbase));  // (sc1: an agent-scope load, as __hip_atomic_load makes it)
}
// The kernel is compiled for MT run loads per window (MT = the most indexed terms of a query of the batch, 2 .. 8; a query of
// fewer terms loads the plane's first bytes for the others -- every load of the loop is unconditional: a conditional one made the
// compiler COPY the arriving registers, i.e. read them before their loads had landed).  R(w + 1) complete: the MT + 2 loads issued
// after it -- P(w), G(w), R(w + 2) -- may still be in flight; G(w): the MT of R(w + 2) behind it; P(w - 1): G(w - 1) and R(w + 1).
#define WN_STR2(x) #x
#define WN_STR(x) WN_STR2(x)
#define WN_OPS1(b) "+v"(b[0])
#define WN_OPS2(b) WN_OPS1(b), "+v"(b[1])
#define WN_OPS3(b) WN_OPS2(b), "+v"(b[2])
#define WN_OPS4(b) WN_OPS3(b), "+v"(b[3])
#define WN_OPS5(b) WN_OPS4(b), "+v"(b[4])
#define WN_OPS6(b) WN_OPS5(b), "+v"(b[5])
#define WN_OPS7(b) WN_OPS6(b), "+v"(b[6])
#define WN_OPS8(b) WN_OPS7(b), "+v"(b[7])
// (issue order at the end of a window w: P(w), the words of its two open passes Ga(w) and Gb(w), then R(w + 2): MT + 3 loads)
template <int MT>
__device__ __forceinline__ void wn_wait_runs(unsigned long long (&b)[MT]) {  // R(w + 1): P, Ga, Gb, R(w + 2) behind it: vmcnt(MT + 3)
    static_assert(MT >= 1 && MT <= 8, "operand lists");
    if constexpr (MT == 1) asm volatile("s_waitcnt vmcnt(4)" : WN_OPS1(b));
    if constexpr (MT == 2) asm volatile("s_waitcnt vmcnt(5)" : WN_OPS2(b));
    if constexpr (MT == 3) asm volatile("s_waitcnt vmcnt(6)" : WN_OPS3(b));
    if constexpr (MT == 4) asm volatile("s_waitcnt vmcnt(7)" : WN_OPS4(b));
    if constexpr (MT == 5) asm volatile("s_waitcnt vmcnt(8)" : WN_OPS5(b));
    if constexpr (MT == 6) asm volatile("s_waitcnt vmcnt(9)" : WN_OPS6(b));
    if constexpr (MT == 7) asm volatile("s_waitcnt vmcnt(10)" : WN_OPS7(b));
    if constexpr (MT == 8) asm volatile("s_waitcnt vmcnt(11)" : WN_OPS8(b));
}
template <int MT>
__device__ __forceinline__ void wn_wait_words(uint32_t &ga, uint32_t &gb) {  // Ga(w) and Gb(w): R(w + 2) behind them: vmcnt(MT)
    if constexpr (MT == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(ga), "+v"(gb));
    if constexpr (MT == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(ga), "+v"(gb));
    if constexpr (MT == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(ga), "+v"(gb));
    if constexpr (MT == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(ga), "+v"(gb));
    if constexpr (MT == 5) asm volatile("s_waitcnt vmcnt(5)" : "+v"(ga), "+v"(gb));
    if constexpr (MT == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(ga), "+v"(gb));
    if constexpr (MT == 7) asm volatile("s_waitcnt vmcnt(7)" : "+v"(ga), "+v"(gb));
    if constexpr (MT == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(ga), "+v"(gb));
}
template <int MT>
__device__ __forceinline__ void wn_wait_theta(unsigned long long &p) {  // P(w - 1): Ga, Gb (w - 1) and R(w + 1) behind it: vmcnt(MT + 2)
    if constexpr (MT == 1) asm volatile("s_waitcnt vmcnt(3)" : "+v"(p));
    if constexpr (MT == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(p));
    if constexpr (MT == 3) asm volatile("s_waitcnt vmcnt(5)" : "+v"(p));
    if constexpr (MT == 4) asm volatile("s_waitcnt vmcnt(6)" : "+v"(p));
    if constexpr (MT == 5) asm volatile("s_waitcnt vmcnt(7)" : "+v"(p));
    if constexpr (MT == 6) asm volatile("s_waitcnt vmcnt(8)" : "+v"(p));
    if constexpr (MT == 7) asm volatile("s_waitcnt vmcnt(9)" : "+v"(p));
    if constexpr (MT == 8) asm volatile("s_waitcnt vmcnt(10)" : "+v"(p));
}
__device__ __forceinline__ void wn_wait_all(uint32_t &g) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(g)); }
// the rare paths inside the loop (a run thicker than one load per lane) load by hand too, and wait for everything: a load the
// compiler knows of under a branch would cost every window its exact waits
__device__ __forceinline__ unsigned long long wn_load_run_now(const uint16_t *addr) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ uint32_t wn_load_u16_now(const uint16_t *addr) {
    uint32_t v;
    asm volatile("global_load_ushort %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(addr));
    return v;
}
// every load issued by hand has landed: their registers are the compiler's again (they stay allocated up to here)
template <int MT>
__device__ __forceinline__ void wn_drain(unsigned long long (&a)[MT], unsigned long long (&b)[MT], uint32_t &g, uint32_t &g2, unsigned long long &p) {
    if constexpr (MT == 1) asm volatile("s_waitcnt vmcnt(0)" : WN_OPS1(a), WN_OPS1(b), "+v"(g), "+v"(g2), "+v"(p));
    if constexpr (MT == 2) asm volatile("s_waitcnt vmcnt(0)" : WN_OPS2(a), WN_OPS2(b), "+v"(g), "+v"(g2), "+v"(p));
    if constexpr (MT == 3) asm volatile("s_waitcnt vmcnt(0)" : WN_OPS3(a), WN_OPS3(b), "+v"(g), "+v"(g2), "+v"(p));
    if constexpr (MT == 4) asm volatile("s_waitcnt vmcnt(0)" : WN_OPS4(a), WN_OPS4(b), "+v"(g), "+v"(g2), "+v"(p));
    if constexpr (MT == 5) asm volatile("s_waitcnt vmcnt(0)" : WN_OPS5(a), WN_OPS5(b), "+v"(g), "+v"(g2), "+v"(p));
    if constexpr (MT == 6) asm volatile("s_waitcnt vmcnt(0)" : WN_OPS6(a), WN_OPS6(b), "+v"(g), "+v"(g2), "+v"(p));
    if constexpr (MT == 7) asm volatile("s_waitcnt vmcnt(0)" : WN_OPS7(a), WN_OPS7(b), "+v"(g), "+v"(g2), "+v"(p));
    if constexpr (MT == 8) asm volatile("s_waitcnt vmcnt(0)" : WN_OPS8(a), WN_OPS8(b), "+v"(g), "+v"(g2), "+v"(p));
}
__device__ __forceinline__ double wave_shl1_f64(double v) {  // lane l gets lane l + 1 (lane 63: itself)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130, 0xf, 0xf, false);  // wave_shl:1
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double((int)hi, (int)lo);
}
// an open pass of second arrivals: what C1 found for the lane's (entry, term), the word requested for it
struct WnPend {
    bool valid, found, task;
    uint32_t x, p, w, gw;
    uint32_t cw;  // the filter word the claim got back (looked at a window later: nobody waits for the atomic where it is issued)
};

// The query's hits from the lists of its items (merge.h's job, done here by the wave that finishes the query's LAST item: round 6 --
// the batched route is one launch, as the one-launch route of scan_range_kernel<..., FUSED> already was; Results::into_sorted_vec,
// search.rs:311-313).  The other waves' lists are read with loads that bypass this CU's vector cache; `map` (the wave's filter,
// free between items: 2048 words) holds (list, entry) of the entries of up to 16 lists, `sv_score` / `sv_doc` (the wave's staged runs)
// the entries at or above the query's final threshold while there are at most 64 of them -- ranked by counting; more than that
// (k > 64, or many ties at the threshold) go through the register top-k.  Leaves the query's share of the per-launch state zero
// for the next launch and returns nothing: the 24-byte records and the count are written where merge_kernel would write them.  A
// query with an item this kernel gave up gets the count NONE32: the host re-runs the batch with scan_many_kernel and merge_kernel
// behind this kernel (vbm25_batch_fetch and the other entry points that hand records to the caller do).
template <int RK>
__device__ __forceinline__ void wn_merge_query(const DevIndex &ix, const uint32_t q, const uint32_t g, const uint32_t k, const uint32_t lane,
                                            uint32_t *map, double *sv_score, uint32_t *sv_doc) {
    const KernArgsP ca = cold_args();  // (every field of the batch is read from the kernarg segment where it is used)
    RegTopK<RK> rtop;
    rtop.init();
    const uint32_t L0 = q * g;
    const unsigned long long theta = __hip_atomic_load(&ca->bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t per_round = k <= 128u ? 16u : 2048u / k;  // (the map holds 2048 entries)
    uint32_t nsv = 0;
    bool ranked = true;
    for (uint32_t lb = 0; lb < g; lb += per_round) {
        uint32_t cnt = 0;
        if (lane < per_round && lb + lane < g) cnt = min(__hip_atomic_load(&ca->bt.res_cnt[L0 + lb + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), k);
        const uint32_t incl = wave_incl_scan_u32(cnt), excl = incl - cnt;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        for (uint32_t j = 0; j < cnt; ++j) map[excl + j] = lane << 16 | j;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t e0 = 0; e0 < total; e0 += 64) {
            bool has = e0 + lane < total;
            double sc = 0;
            uint32_t d = 0;
            if (has) {
                const uint32_t ds = map[e0 + lane];
                const size_t at = (size_t)(L0 + lb + (ds >> 16)) * k + (ds & 0xffffu);
                sc = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(&ca->bt.res_score[at]), __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_AGENT));
                d = __hip_atomic_load(&ca->bt.res_doc[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                has = (unsigned long long)__double_as_longlong(sc) >= theta;
            }
            const unsigned long long hm = __ballot(has);
            if (ranked && nsv + (uint32_t)__popcll(hm) <= 64u) {
                if (has) {
                    const uint32_t at = nsv + wn_mbcnt(hm);
                    sv_score[at] = sc;
                    sv_doc[at] = d;
                }
                nsv += (uint32_t)__popcll(hm);
                continue;
            }
            if (ranked) {  // more than 64 survivors: the buffered ones first, then everything through the register top-k
                ranked = false;
                __builtin_amdgcn_wave_barrier();
                const bool hb = lane < nsv;
                rtop.offer(hb, hb ? sv_score[lane] : 0.0, hb ? sv_doc[lane] : 0u, k, lane);
            }
            rtop.offer(has, sc, d, k, lane);  // (the items' document ranges are disjoint: no document comes twice)
        }
        __builtin_amdgcn_wave_barrier();
    }
    auto emit = [&](uint32_t i, double sc, uint32_t d) {
        // 24-byte record written as three 64-bit words so that padding bytes are zero
        const uint16_t *pl = ix.doc_payload + 3ull * d;
        unsigned long long *out = reinterpret_cast<unsigned long long *>(ca->bt.hits + (size_t)q * k + i);
        out[0] = (unsigned long long)__double_as_longlong(sc);
        out[1] = (unsigned long long)d | (unsigned long long)pl[0] << 32 | (unsigned long long)pl[1] << 48;
        out[2] = (unsigned long long)pl[2];
    };
    uint32_t n;
    if (ranked) {
        // at most 64 survivors, one per lane: an entry's place = the number of entries better than it (score descending, ties by
        // ascending document)
        __builtin_amdgcn_wave_barrier();
        const bool mine = lane < nsv;
        const double sc = mine ? sv_score[lane] : 0.0;
        const uint32_t d = mine ? sv_doc[lane] : 0u;
        uint32_t place = 0;
        for (uint32_t j = 0; j < nsv; ++j) {
            const double sj = readlane_f64(sc, j);
            const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)d, (int)j);
            place += better(sj, dj, sc, d) ? 1u : 0u;
        }
        n = min(nsv, k);
        if (mine && place < k) emit(place, sc, d);
    } else {
        n = rtop.cnt;
#pragma unroll
        for (int r = 0; r < RK; ++r)
            if (r * 64 + lane < n) emit(r * 64 + lane, rtop.score[r], rtop.doc[r]);
    }
    // the query's share of the per-launch state back to zero (the next launch starts on it); what the test aids want to see
    // afterwards is kept aside
    uint32_t nf = 0;
    for (uint32_t i = lane; i < g; i += 64) {
        nf += __hip_atomic_load(&ca->bt.item_failed[L0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1u : 0u;
        __hip_atomic_store(&ca->bt.item_failed[L0 + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ca->bt.res_cnt[L0 + i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nf += __shfl_xor(nf, o);
    if (lane == 0) {
        ca->bt.n_hits[q] = nf ? NONE32 : n;
        ca->bt.q_failed[q] = nf;
        ca->bt.theta_last[q] = theta;
        __hip_atomic_store(&ca->bt.theta[q], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ca->bt.fused_state[1 + q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// RK: rows of 64 entries of the wave's top-k in registers (k <= 64 RK)
// MT: the most indexed terms of a query of the batch (2 .. 8).  A query of fewer terms gets NULL terms for the rest -- the window table
// at the start of win_off, all zeros: runs without postings -- so that the per-term tests of the window loop are decided at compile
// time and the terms' marks are one straight-line block (with the number of terms a run-time value: 0.215 instead of 0.199 ms on C3)
template <int MT, int RK>
__global__ void __launch_bounds__(wn_waves(MT, RK) * 64, (wn_waves(MT, RK) + 3) / 4) scan_win_kernel(DevIndex ix, DevBatch bt) {
    constexpr int WN_WAVES = wn_waves(MT, RK), WN_WG = WN_WAVES * 64;
    static_assert(sizeof(WinWave<MT>) + WN_BM_WORDS * 4 == wn_wave_lds(MT), "wn_waves() knows the size");
    __shared__ alignas(8192) uint32_t BM[WN_WAVES * WN_BM_WORDS];  // the waves' filters, 2^16 bits each
    __shared__ WinWave<MT> SW[WN_WAVES];
    __shared__ double S1[256];  // k1 (1 - b + b len(f) / avgdl) per fieldnorm (bm25.rs:349-352)
    const uint32_t lane = threadIdx.x & 63;
    WinWave<MT> &S = SW[uni(threadIdx.x >> 6)];
    uint32_t *const bm = &BM[uni(threadIdx.x >> 6) * WN_BM_WORDS];
    WnBm bmbase;
    bmbase.base = uni((uint32_t)(uintptr_t)(wn_lds_u32 *)bm);
    bmbase.mask = 0x1ffcu;
    asm volatile("" : "+v"(bmbase.mask));
    const uint32_t k = bt.k, g = bt.win_g, n_items = bt.nq * g, NWIN = ix.n_win;
#ifdef VBM25_DEV
    const uint32_t dbg = bt.win_dbg;  // timing experiments (wrong results): 1 no cold pass, 2 no completion, 4 no second arrivals, 32 no marks, 64 no wipe
#else
    constexpr uint32_t dbg = 0;       // (the product has no switch that changes results: the experiments need make libvbm25_dev.so)
#endif
    const uint16_t *ids16 = reinterpret_cast<const uint16_t *>(ix.post_id16);
    for (uint32_t i = threadIdx.x; i < 256u; i += WN_WG) S1[i] = ix.s1[i];
#pragma unroll
    for (int i = 0; i < WN_BM_WORDS / 4 / 64; ++i) reinterpret_cast<uint4 *>(bm)[lane + 64 * i] = make_uint4(0, 0, 0, 0);
    __syncthreads();  // (the table; the only workgroup barrier of the kernel -- from here on the waves go their own ways)
#ifdef VBM25_PROFILE
    unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif

    // A wave's FIRST item is its own number (thousands of waves drawing from one counter at kernel start queued at that address for
    // up to 33 us of a 270 us launch); further items -- only when the launch has more items than waves -- come from the counter, drawn
    // at the END of the item before (the round trip hides behind the cold pass).  (The development switch 8 draws the first one from
    // the counter too.)
    const uint32_t n_waves = gridDim.x * (uint32_t)WN_WAVES;
    uint32_t drawn = blockIdx.x * (uint32_t)WN_WAVES + uni(threadIdx.x >> 6);
    uint32_t merge_q = NONE32;  // (win_fuse: the query this wave finished the last item of)
    if (dbg & 8u) {
        uint32_t d0 = 0;
        if (lane == 0) d0 = atomicAdd(cold_args()->bt.work_ctr, 1u);
        drawn = uni(d0);
    }
    for (;;) {
        uint32_t item = drawn;
        if (item >= n_items) break;
        uint32_t w_lo, w_hi;
        {
            const KernArgsP ca = cold_args();
            const uint32_t oo = ca->bt.order_on;
            if (oo == 1u) item = uni(ca->bt.item_order[item]);  // the host's order: longest items first
            else if (oo == 2u) {
                // the skewed layout of a batch whose queries kept their order (C3), by arithmetic: a workgroup takes four queries, part k of
                // them goes to the slots 4 k .. 4 k + 3 (search.hip, set_queries: the same formula fills item_order) -- one dependent
                // round trip to memory less in every item's setup
                const uint32_t full12 = (ca->bt.nq >> 2) * 12u;
                if (item < full12) {
                    const uint32_t wgi = item / 12u, r = item - wgi * 12u;
                    item = (wgi * 4u + (r & 3u)) * 3u + (r >> 2);
                }
            }
        }
        PROF_T(t_item);
        const uint32_t q = item / g, part = item - q * g;
        {
            const KernArgsP ca = cold_args();
            if (g <= 16u && ca->bt.win_cut[g] != 0u) {
                w_lo = ca->bt.win_cut[part];
                w_hi = ca->bt.win_cut[part + 1u];
            } else {
                w_lo = (uint32_t)((unsigned long long)NWIN * part / g);
                w_hi = (uint32_t)((unsigned long long)NWIN * (part + 1u) / g);
            }
        }

        // ---- item setup: lane t = term t of the query (ascending key order)
        uint32_t m = 0, term = NONE32;
        uint32_t fbi = NONE32;  // the term's first block in the plane of ids: its own first block, or (an index without post_id16) its place in the batch's scratch plane
        {
            const KernArgsP ca = cold_args();
            // (the host sends queries of <= 64 terms this way; when they all have the same number of terms nobody waits for q_off)
            const uint32_t qs = ca->bt.q_stride;
            const uint32_t qb = qs ? qs * q : uni(ca->bt.q_off[q]), qe = qs ? qs * (q + 1u) : uni(ca->bt.q_off[q + 1]);
            const uint32_t *idfb = ca->bt.id16_fb;
            const uint32_t tt = lane < qe - qb ? ca->bt.term_ids[qb + lane] : NONE32;
            const uint32_t ti = idfb && lane < qe - qb ? idfb[qb + lane] : NONE32;
            const bool ok = tt < ix.n_terms;  // search.rs:59-61
            const unsigned long long okm = __ballot(ok);
            if (ok) {
                S.list[wn_mbcnt(okm)] = tt;
                if (wn_mbcnt(okm) < 32u) S.list[32u + wn_mbcnt(okm)] = ti;  // (queries of <= 64 terms come this way; the kernel takes the ones of <= 8)
            }
            m = (uint32_t)__popcll(okm);
            __builtin_amdgcn_wave_barrier();
            if (lane < min(m, (uint32_t)MT)) {
                term = S.list[lane];
                fbi = S.list[32u + lane];
            }
            __builtin_amdgcn_wave_barrier();
        }
        const bool act = lane < min(m, (uint32_t)MT);
        uint32_t fb = 0, wb = 0;
        double s0 = 0.0, kth = 0.0;
        {
            const KernArgsP ca = cold_args();
            const double *kub = ca->ix.term_kth_ub;
            if (act) {
                fb = ca->ix.term_first_block[term];
                if (fbi == NONE32) fbi = fb;
                s0 = ca->ix.term_s0[term];
                wb = ca->ix.term_win[term];
                if (kub) {  // (the smallest 2^i >= k: at least k documents of the term score that much)
                    uint32_t kidx = 0;
                    while ((1u << kidx) < k) ++kidx;
                    kth = kub[(size_t)term * 9 + kidx];
                }
            }
        }
        // (a lane without a term -- the null terms of a shorter query -- keeps block 0: every address built from it stays inside the plane;
        // NONE32 there sent the unconditional loads of the rare search beyond the staged run to a wild address in batches of mixed lengths)
#ifndef VBM25_EXPERIMENT_WILD_NULL_TERMS  // (tools/win_variant.sh: the regression test of this line must fail on that build)
        if (!act) fbi = 0u;
#endif
        // (the host routes only queries of <= WN_T indexed terms that all have a table this way)
        bool failed = m > (uint32_t)MT || __ballot(act && wb == NONE32) != 0ull;
        if (failed) m = 0;
        constexpr uint32_t mm = (uint32_t)MT;  // (terms beyond the query's own, and every term of an item given up here: the null table)
        const uint32_t wbs = act && !failed ? wb : 0u;  // (lanes without a term: the null table at win_off[0 ..]; every load below is unconditional)
        // the term's postings as bytes (lane = term), and per term the numbers of its postings below the item's window boundaries
        // (lane i = boundary w_lo + i; the host cuts items of at most 63 windows): no boundary is loaded inside the window loop
        const unsigned long long pbase = (unsigned long long)ids16 + (act && !failed ? 256ull * fbi : 0ull);
        const uint32_t nw = w_hi - w_lo;
        failed = failed || nw > 63u;
        uint32_t wo[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
            wo[t] = ix.win_off[(uint32_t)__builtin_amdgcn_readlane((int)wbs, t) + min(w_lo + lane, NWIN)];
        uint32_t wS = 0, wE = 0;  // lane = term: its postings below the item's two ends
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            if (lane == (uint32_t)t) {
                wS = (uint32_t)__builtin_amdgcn_readlane((int)wo[t], 0);
                wE = (uint32_t)__builtin_amdgcn_readlane((int)wo[t], (int)min(nw, 63u));
            }
        }
        uint32_t wA = 0, wB = wS;  // the boundaries of the window being worked on (its upper ones written with its marks): of the term of the lane's (entry, term) pair in a completion pass
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kth = fmax(kth, __shfl_xor(kth, o));
        unsigned long long th = (unsigned long long)__double_as_longlong(kth);  // theta0
        {
            unsigned long long pg = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pg = ((unsigned long long)uni((uint32_t)(pg >> 32)) << 32) | uni((uint32_t)pg);
            if (lane == 0 && th > pg) atomicMax(&bt.theta[q], th);
            if (pg > th) th = pg;
        }
        if (lane == 0) {
            // the records plan_kernel would have made: scan_many_kernel (items this kernel gives up) reads them
            const KernArgsP ca = cold_args();
            Item rec;
            rec.q = q;
            rec.doc_lo = w_lo << 16;
            rec.doc_hi = (uint32_t)min((unsigned long long)w_hi << 16, (unsigned long long)ix.n_docs);
            rec.m = m;
            ca->bt.items[item] = rec;
            if (item == 0) *ca->bt.n_items = n_items;
            if (part == 0) ca->bt.q_item_base[q] = item;
            if (item + 1 == n_items) ca->bt.q_item_base[q + 1] = n_items;
        }

        RegTopK<RK> rtop;
        rtop.init();
        unsigned long long published = 0;
        auto offer = [&](bool has, double sc, uint32_t d) {
            has = has && (unsigned long long)__double_as_longlong(sc) >= th && (rtop.cnt < k || better(sc, d, rtop.kth_s, rtop.kth_d));
            if (!__ballot(has)) return;
            rtop.offer(has, sc, d, k, lane);
            if (rtop.cnt >= k) {
                const unsigned long long kb = (unsigned long long)__double_as_longlong(rtop.kth_s);
                if (kb > th) th = kb;
            }
        };

        // the run of every term in window w_lo + i, four postings per lane: R(w_lo + i)
        const uint32_t lane8 = 8u * lane, lane4 = 4u * lane;
        auto load_runs = [&](unsigned long long (&dst)[MT], const uint32_t i) {
            const uint32_t ic = min(i, 63u);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                // (exactly MT loads, whatever the query's number of terms: the waits count them)
                const unsigned long long a = (((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pbase >> 32), t) << 32) |
                                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pbase, t)) +
                                             2ull * ((uint32_t)__builtin_amdgcn_readlane((int)wo[t], (int)ic) & ~3u);
                wn_load_run(dst[t], lane8, a);
            }
        };
        WnPend pa{}, pb{};  // the two open passes of the last window (entries 0 .. epp - 1 and epp .. 2 epp - 1)
        unsigned long long pgv = 0;  // the query's shared threshold, polled a window ahead
        uint32_t vzero = 0;
        asm volatile("" : "+v"(vzero));
        // the loop's steady state from its first window on: R(w_lo), then a P and a G, then R(w_lo + 1) -- ten loads behind R(w_lo)
        unsigned long long bufa[MT], bufb[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) bufa[t] = bufb[t] = 0;
        load_runs(bufa, 0u);
        wn_load_theta(pgv, vzero, &bt.theta[q]);
        wn_load_word(pa.gw, ix.post_tfn);
        wn_load_word(pb.gw, ix.post_tfn);
        load_runs(bufb, 1u);
        // (landed before the loop is entered: the compiler may move these registers on the way in -- a copy of a register whose
        // load is still in flight would copy what was there before.  One exposed round trip per item; inside the loop a buffer is
        // written by its loads and read by its marks only)
        wn_drain<MT>(bufa, bufb, pa.gw, pb.gw, pgv);

        // ---- completion of the second arrivals in two halves: C1 (after a window's marks) finds the postings -- lane = (entry, term),
        // bisection of the term's staged run -- and requests their tf / fieldnorm words; C2 scores, sums and offers.  The first
        // pass of a window is finished only after the NEXT window's marks: the round trip to HBM hides behind them.
        const uint32_t inv_m = mm ? (65536u + mm - 1u) / mm : 0u, epp = mm ? 64u / mm : 0u;
        const uint32_t el = (lane * inv_m) >> 16, tl = mm ? lane - el * mm : 0u;  // the lane's entry of a pass and its term
        const uint32_t fbl = (uint32_t)__shfl((int)fb, (int)tl);
        const uint32_t fbil = (uint32_t)__shfl((int)fbi, (int)tl);  // (the same block unless the ids come from the batch's scratch plane)
        const double s0l = __shfl(s0, (int)tl);
        wB = (uint32_t)__shfl((int)wS, (int)tl);
        bool notf = false;
        // C1 in two halves: the search (one straight-line block: the entry, its id, the 4-ary / binary search of the term's staged run) and
        // the rest (the rare search in memory beyond the staged part, the claim, the pass's record)
        struct C1S {
            bool task, found, more;
            uint32_t x, base, plo, len, slen;
        };
        // (`hook(step)` runs between the search's round trips to the LDS: the merged block hands the steps of C2's f64 divide in)
        auto c1_search = [&](const uint32_t e0, const uint32_t nd, auto &&hook) -> C1S {
            C1S r;
            const bool task = el < epp && e0 + el < nd;
            const uint32_t ent = S.list[task ? e0 + el : 0u];
            // the entry's id: of a slot (group, term of the group, lane, posting of the lane) of the staged runs, or the id itself (bit 31)
            const uint32_t esl = ent & 31u;
            const uint32_t xs = S.stage[(task && !(ent >> 31) ? ((esl >> 2) + 5u * ((ent >> 11) & 1u)) * (uint32_t)WN_SLOT + ((ent >> 5) & 63u) * 4u + (esl & 3u) : 0u)];
            const uint32_t x = (ent >> 31) ? ent & 0xffffu : xs;
            const uint32_t plo = wA, phi = wB;
            const uint32_t pal = plo & ~3u, len = phi - plo;
            const uint16_t *sb = &S.stage[tl * WN_SLOT + (plo - pal)];
            uint32_t base = 0;  // the last posting of the run whose id is <= x
            const uint32_t slen = min(len, (uint32_t)WN_SLOT - (plo - pal));  // the staged part of the run
            uint32_t n2 = task ? slen : 0u;
#define WN_BISECT_STEP(IT)                                              \
    {                                                                   \
        const uint32_t half = n2 >> 1;                                  \
        const uint32_t probe = sb[base + half];                         \
        hook(std::integral_constant<int, IT>());                        \
        if (probe <= x) base += half;                                   \
        n2 -= half;                                                     \
    }
            // (an interpolated start -- posting x slen / 2^16 and six steps over the 64 postings around it, the whole run for the lanes whose
            // search ends unproven at an edge -- was measured: 0.1763 -> 0.1824 ms on C3: the guess, the edge test and the branch cost more
            // than the two round trips to the LDS they save, as the 4-ary search did)
            WN_BISECT_STEP(0) WN_BISECT_STEP(1) WN_BISECT_STEP(2) WN_BISECT_STEP(3) WN_BISECT_STEP(4) WN_BISECT_STEP(5) WN_BISECT_STEP(6) WN_BISECT_STEP(7)
#undef WN_BISECT_STEP
            // (both reads unconditional: a read under a lane condition is a branch, and a branch ends the block the scheduler works in)
            uint32_t v_at = sb[base], v_last = sb[slen ? slen - 1u : 0u];
            asm volatile("" : "+v"(v_at), "+v"(v_last));  // (opaque: the compiler sinks a load whose only use is conditional into the condition)
            r.found = task && len != 0u && v_at == x;
            // a run thicker than its stage row and a document beyond the staged part: the rest of the run, in memory
            r.more = task && !r.found && slen < len && v_last < x;
            r.task = task;
            r.x = x;
            r.base = base;
            r.plo = plo;
            r.len = len;
            r.slen = slen;
            return r;
        };
        auto c1_finish = [&](WnPend &d, const C1S &r, const uint32_t w) {
            bool found = r.found;
            uint32_t base = r.base;
            bool more = r.more;
            const uint32_t x = r.x;
            if (__ballot(more)) {
                const uint16_t *gb = ids16 + 128ull * fbil + r.plo;
                uint32_t lo = r.slen, hi = r.len;
                while (__ballot(more && lo < hi)) {  // first posting of [slen, len) with id >= x
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint32_t v = wn_load_u16_now(gb + (more ? mid : 0u));
                    if (more && lo < hi) {
                        if (v < x) lo = mid + 1u;
                        else hi = mid;
                    }
                }
                more = more && lo < r.len;
                const uint32_t v = wn_load_u16_now(gb + (more ? lo : 0u));
                if (more && v == x) {
                    found = true;
                    base = lo;
                }
            }
            // the document is offered by the first of its entries that claims it: the bit goes, whoever saw it there owns the document
            uint32_t cw = 0;
            if (r.task && tl == 0u) {
                const uint32_t bit = 1u << (x & 31u);
                cw = __hip_atomic_fetch_and((wn_lds_u32 *)(((x >> 3) & bmbase.mask) | bmbase.base), ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            d.task = r.task;
            d.found = found;
            d.cw = cw;
            d.x = x;
            d.p = r.plo + base;
            d.w = w;
        };
        auto c1 = [&](WnPend &d, const uint32_t e0, const uint32_t nd, const uint32_t w) {
            const C1S r = c1_search(e0, nd, [](auto) {});
            c1_finish(d, r, w);
        };
        // the word of post_tfn that holds the posting C1 found (word 0 for the lanes that found none: the load is unconditional)
        auto c1_request = [&](WnPend &d) { wn_load_word(d.gw, ix.post_tfn + (d.found ? 64ull * fbl + (d.p >> 1) : 0ull)); };
        // C2 in two halves as well: the arithmetic (straight-line) and the offer
        struct C2R {
            bool okd;
            double acc;
            uint32_t doc;
        };
        auto c2_calc = [&](const WnPend &d) -> C2R {
            // (every lane computes -- no branch around the divide, so that it can be scheduled into the search's waits; a lane that found
            // nothing divides by S1[..] + 0 > 0 and drops the result)
            double c;
            {
                const uint32_t ww = d.gw >> ((d.p & 1u) * 8u);
                const uint32_t tfv = d.found ? ww & 0xffu : 0u, fn = (ww >> 16) & 0xffu;
                notf = notf || (d.found && tfv == 0u);  // (a term frequency above 255: the word holds zeros -- not this kernel's item)
                const double tf = (double)tfv;
                c = (tf * s0l) / (tf + S1[fn]);  // Cache::evaluate, bm25.rs:355-358 (tf = 0: + 0.0)
            }
            const unsigned long long fm = __ballot(d.found);
            C2R r;
            r.okd = false;
            r.acc = 0.0;
            r.doc = d.w << 16 | d.x;
            {   // the row's sum in ascending key order (evaluate.rs:43-72) at the row's first lane: the neighbours' values come by wave shifts
                double sh = c;
                r.acc = c;
#pragma unroll
                for (int t = 1; t < MT; ++t) {
                    sh = wave_shl1_f64(sh);
                    r.acc += sh;  // absent terms add 0.0
                }
            }
            if (d.task && tl == 0u) {
                const uint32_t my = (uint32_t)(fm >> lane) & ((1u << mm) - 1u);
                // the entry that claimed the document completes it (one offer per document) -- if two lists hold it
                r.okd = ((d.cw >> (d.x & 31u)) & 1u) != 0u && (my & (my - 1u)) != 0u;
            }
            return r;
        };
        auto c2 = [&](const WnPend &d) {
            const C2R r = c2_calc(d);
            offer(r.okd, r.acc, r.doc);
        };
        // C2 with its divide taken apart (the merged block): what LLVM makes of an f64 `/` -- div_scale, rcp, two Newton steps,
        // div_fmas, div_fixup: the correctly rounded quotient, bit for bit what `/` gives -- as eight steps that the search's hook runs
        // one per round trip to the LDS
        struct C2Div {
            double num, den, sc0, rcp, f0, f1, f2, sc1, f3, mul, f4, q;
            bool vcc;
        };
        auto c2_div_begin = [&](const WnPend &d, C2Div &D) {
            const uint32_t ww = d.gw >> ((d.p & 1u) * 8u);
            const uint32_t tfv = d.found ? ww & 0xffu : 0u, fn = (ww >> 16) & 0xffu;
            notf = notf || (d.found && tfv == 0u);
            const double tf = (double)tfv;
            D.num = tf * s0l;       // (a lane that found nothing: 0 / S1[..] = + 0.0)
            D.den = tf + S1[fn];
        };
        auto c2_div_step = [&](C2Div &D, auto itc) {
            constexpr int it = decltype(itc)::value;
            if constexpr (it == 0) {
                bool unused;
                D.sc0 = __builtin_amdgcn_div_scale(D.num, D.den, false, &unused);
                D.rcp = __builtin_amdgcn_rcp(D.sc0);
            }
            if constexpr (it == 1) D.f0 = __builtin_fma(-D.sc0, D.rcp, 1.0);
            if constexpr (it == 2) D.f1 = __builtin_fma(D.rcp, D.f0, D.rcp);
            if constexpr (it == 3) D.f2 = __builtin_fma(-D.sc0, D.f1, 1.0);
            if constexpr (it == 4) {
                D.sc1 = __builtin_amdgcn_div_scale(D.num, D.den, true, &D.vcc);
                D.f3 = __builtin_fma(D.f1, D.f2, D.f1);
            }
            if constexpr (it == 5) D.mul = D.sc1 * D.f3;
            if constexpr (it == 6) D.f4 = __builtin_fma(-D.sc0, D.mul, D.sc1);
            if constexpr (it == 7) D.q = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(D.f4, D.f3, D.mul, D.vcc), D.den, D.num);
        };
        auto c2_sum = [&](const WnPend &d, const double c) -> C2R {
            const unsigned long long fm = __ballot(d.found);
            C2R r;
            r.okd = false;
            r.doc = d.w << 16 | d.x;
            double sh = c;
            r.acc = c;
#pragma unroll
            for (int t = 1; t < MT; ++t) {
                sh = wave_shl1_f64(sh);
                r.acc += sh;  // absent terms add 0.0
            }
            if (d.task && tl == 0u) {
                const uint32_t my = (uint32_t)(fm >> lane) & ((1u << mm) - 1u);
                r.okd = ((d.cw >> (d.x & 31u)) & 1u) != 0u && (my & (my - 1u)) != 0u;
            }
            return r;
        };

        // One window: `cur` holds its runs (requested two windows ago) and takes the runs of the window after next at the end.  Two
        // buffers in fixed roles, the loop below alternates them: a run is loaded into the very registers it is marked from (a
        // rotation of the buffers made the compiler copy every arriving run, one load at a time).  False: the item is given up.
        auto window = [&](unsigned long long (&cur)[MT], const uint32_t w) -> bool {
            // ---- marks of window w (its runs arrived while the last window was worked on).  Two phases per group of five pairs: ALL
            // the group's atomics are issued before the first returned word is looked at -- the LDS executes a wave's operations in
            // order, so every returned word already reflects the marks issued before it, and the group costs one round trip to the
            // LDS instead of five.  The last window's open pass of second arrivals (C2) is finished between the two phases.
            uint32_t nd = 0;
            wA = wB;  // (the last window's upper boundaries)
            PROF_T(t_0);
            wn_wait_runs<MT>(cur);
            PROF_T(t_1);
            PROF_ADD(1, t_0, t_1);
#ifdef VBM25_PROFILE
            unsigned long long t_p1 = t_1;
#endif
            // Groups of <= 5 terms.  Phase 1: the group's runs are staged and marked -- a lane marks ALL FOUR of its postings when its
            // first one lies below the run's end, with no test per posting: the <= 3 postings before the run's first (the last window's,
            // in lane 0) and behind its last (the next window's, in the run's last lane) set bits that belong to nobody (PHANTOMS, about
            // three per term and window among 217 real marks).  A phantom can only ADD entries to the list of second arrivals -- a
            // real posting that finds a phantom's bit, a phantom that finds a real one -- never hide one: every document with postings
            // in two lists still has a real mark that finds the bit set.  What an entry is worth is decided by the completion, which
            // finds the document's postings in the staged runs themselves; a document is offered by the FIRST of its entries to CLAIM
            // it (C1 clears its bit with a returning atomic), so duplicates of any origin are harmless.  Phase 2: the returned words
            // -> one flag per posting (bit x of the word that came back) -> the flagged slots appended to the list, a slot per lane
            // and round (a lane rarely holds two).
#pragma unroll
            for (int gq = 0; gq < (MT > 5 ? 2 : 1); ++gq) {
                constexpr int GT = 5;
                uint32_t ho[GT][4];  // the words that came back (terms after the window's first)
                uint32_t en[GT];     // postings from the aligned start of the run's first load to the run's end: lanes with 4 lane < en mark
#pragma unroll
                for (int u = 0; u < GT; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("; ho undefined" : "=v"(ho[u][j]));
                if (!(dbg & 128u)) {
#pragma unroll
                    for (int u = 0; u < GT; ++u) {
                        const int t = GT * gq + u;
                        en[u] = 0;
                        if (t < MT) {
                            const uint32_t o_lo = (uint32_t)__builtin_amdgcn_readlane((int)wo[t < MT ? t : 0], (int)(w - w_lo));
                            const uint32_t o_hi = (uint32_t)__builtin_amdgcn_readlane((int)wo[t < MT ? t : 0], (int)(w - w_lo + 1u));
                            const uint32_t o_al = o_lo & ~3u;
                            en[u] = o_hi - o_al;
                            wB = tl == (uint32_t)t ? o_hi : wB;  // (lane = (entry, term) of a completion pass: its term's boundary)
                            const uint2 run = make_uint2((uint32_t)cur[t < MT ? t : 0], (uint32_t)(cur[t < MT ? t : 0] >> 32));
                            *reinterpret_cast<uint2 *>(&S.stage[t * WN_SLOT + 4u * lane]) = run;
                            if (lane4 < en[u] && !(dbg & 32u)) {
                                const uint32_t x0 = run.x & 0xffffu, x1 = run.x >> 16, x2 = run.y & 0xffffu, x3 = run.y >> 16;
                                if (t == 0) {  // the window's first term: the filter is empty, nothing can be there yet
                                    wn_or(bmbase, x0, 1u << (x0 & 31u));
                                    wn_or(bmbase, x1, 1u << (x1 & 31u));
                                    wn_or(bmbase, x2, 1u << (x2 & 31u));
                                    wn_or(bmbase, x3, 1u << (x3 & 31u));
                                } else if (t == MT - 1) {  // the window's last term only LOOKS: nobody comes after it to find its bits
                                    ho[u][0] = *(wn_lds_u32 *)(((x0 >> 3) & bmbase.mask) | bmbase.base);
                                    ho[u][1] = *(wn_lds_u32 *)(((x1 >> 3) & bmbase.mask) | bmbase.base);
                                    ho[u][2] = *(wn_lds_u32 *)(((x2 >> 3) & bmbase.mask) | bmbase.base);
                                    ho[u][3] = *(wn_lds_u32 *)(((x3 >> 3) & bmbase.mask) | bmbase.base);
                                } else {
                                    ho[u][0] = wn_or_rtn(bmbase, x0, 1u << (x0 & 31u));
                                    ho[u][1] = wn_or_rtn(bmbase, x1, 1u << (x1 & 31u));
                                    ho[u][2] = wn_or_rtn(bmbase, x2, 1u << (x2 & 31u));
                                    ho[u][3] = wn_or_rtn(bmbase, x3, 1u << (x3 & 31u));
                                }
                            }
                            if (en[u] > (uint32_t)WN_SLOT) {  // a run that one load per lane does not hold: the rest, not staged (C1 reads it from memory)
                                const uint16_t *rest = ids16 + 128ull * (uint32_t)__builtin_amdgcn_readlane((int)fbi, t) + 4u * lane;
                                const uint32_t n = o_hi - o_lo;
#pragma nounroll
                                for (uint32_t o = o_al + (uint32_t)WN_SLOT; o < o_hi; o += (uint32_t)WN_SLOT) {
                                    const unsigned long long mv = wn_load_run_now(rest + o);
                                    const uint2 more = make_uint2((uint32_t)mv, (uint32_t)(mv >> 32));
                                    if (t == 0) wn_pair<false, MT>(S, bmbase, more, o + 4u * lane - o_lo, n, (uint32_t)t, nd);
                                    else wn_pair<true, MT>(S, bmbase, more, o + 4u * lane - o_lo, n, (uint32_t)t, nd);
                                }
                            }
                        }
                    }
                }
#ifdef VBM25_PROFILE
                if (gq == 0) {
                    const unsigned long long t_2 = __builtin_readcyclecounter();
                    prof[2] += t_2 - t_1;
                    t_p1 = t_2;
                }
#endif
                if (!(dbg & (128u | 32u))) {
                    uint32_t flags = 0;  // bit 4 u + j: the lane's posting j of the group's term u found its bit set
#pragma unroll
                    for (int u = 0; u < GT; ++u) {
                        const int t = GT * gq + u;
                        if (t >= 1 && t < MT) {
                            if (lane4 < en[u]) {
                                const uint32_t v0 = (uint32_t)cur[t < MT ? t : 0], v1 = (uint32_t)(cur[t < MT ? t : 0] >> 32);
                                uint32_t f = (ho[u][0] >> (v0 & 31u)) << 31;  // (the four flags shifted in from the top, one instruction each)
                                f = __builtin_amdgcn_alignbit(ho[u][1] >> ((v0 >> 16) & 31u), f, 1);
                                f = __builtin_amdgcn_alignbit(ho[u][2] >> (v1 & 31u), f, 1);
                                f = __builtin_amdgcn_alignbit(ho[u][3] >> ((v1 >> 16) & 31u), f, 1);
                                flags |= (f >> 28) << (4 * u);
                            }
                        }
                    }
                    while (__ballot(flags != 0u)) {
                        const bool has = flags != 0u;
                        const uint32_t sl = (uint32_t)__ffs((int)flags) - 1u;
                        flags &= flags - 1u;
                        const unsigned long long mk = __ballot(has);
                        const uint32_t pos = nd + wn_mbcnt(mk);
                        if (has && pos < (uint32_t)WN_LIST) S.list[pos] = sl | lane << 5 | (uint32_t)gq << 11;  // (the slot: C1 reads the id from the staged run)
                        nd += (uint32_t)__popcll(mk);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            PROF_T(t_4);
#ifdef VBM25_PROFILE
            prof[3] += t_4 - t_p1;
#endif
            // more second arrivals than the list holds: the lists are too dense here for this kernel
            if (nd > (uint32_t)WN_LIST) failed = true;
#ifdef VBM25_PROFILE
            prof[0] += 1;
            prof[10] += nd;
#endif
            if (dbg & 4u) nd = 0;

            // ---- the last window's open pass of second arrivals: its tf / fieldnorm words were requested a whole window ago (between the
            // two phases of the marks, right behind the request, the wait for them was still exposed)
            PROF_T(t_w0);
            PROF_ADD(4, t_4, t_w0);
            wn_wait_words<MT>(pa.gw, pb.gw);
            PROF_T(t_w1);
            PROF_ADD(5, t_w0, t_w1);
#ifdef VBM25_PROFILE
            unsigned long long t_w2 = 0;
#endif
            // ONE straight-line block: the arithmetic of the last window's first pass and the search of this window's first pass are
            // independent chains (an f64 divide; eleven round trips to the LDS) -- in one block the scheduler fills the one's waits
            // with the other's instructions.  Both run unconditionally: an invalid pass has found nothing, an empty list gives no task.
            {
                const uint32_t nd1 = !failed && !(dbg & 2u) ? nd : 0u;
                C2Div dv;
                c2_div_begin(pa, dv);
                const C1S sa = c1_search(0u, nd1, [&](auto itc) { c2_div_step(dv, itc); });
                const C2R ra = c2_sum(pa, dv.q);
                offer(ra.okd, ra.acc, ra.doc);
                if (pb.valid) c2(pb);
                pb.valid = false;
                pa.found = pb.found = false;
                PROF_MARK(t_w2);
                PROF_ADD(6, t_w1, t_w2);
                c1_finish(pa, sa, w);
                pa.valid = nd1 != 0u;
                if (nd1 > epp) {
                    for (uint32_t e0 = ((nd1 - 1u) / epp) * epp; e0 >= 2u * epp; e0 -= epp) {
                        c1(pb, e0, nd1, w);
                        c1_request(pb);
                        wn_wait_all(pb.gw);
                        c2(pb);
                    }
                    pb.found = false;
                    c1(pb, epp, nd1, w);
                    pb.valid = true;
                }
            }
            // (the wipe: behind the claims, which read the filter)
#ifdef VBM25_WIN_WIPE_STORES  // (the comparison: tools/win_variant.sh oldwipe -DVBM25_WIN_WIPE_STORES)
            if (!(dbg & 64u))
#pragma unroll
                for (int i = 0; i < WN_BM_WORDS / 4 / 64; ++i) reinterpret_cast<uint4 *>(bm)[lane + 64 * i] = make_uint4(0, 0, 0, 0);
#else
            if (!(dbg & 64u)) {
                // 32 stores of 256 bytes whose address is M0 + a 16-bit offset + 4 lane (ds_write_addtid_b32: no address register to move,
                // two LDS cycles each -- MI355X_MICROARCH.md, LDS) instead of eight 16-byte stores per lane (13 cycles each): the LDS pipe
                // is what a window waits for.  M0 is used with all its bits -- the filters of the workgroup's later waves lie beyond
                // 64 KB; tools/ubench/addtid_probe.hip, profiles/r6_addtid_probe.txt -- and an LDS add-TID instruction must not follow the
                // SALU write of M0 directly (one wait state, which the compiler cannot insert inside an asm statement: without the s_nop the
                // wave's FIRST wipe wrote its first 256 bytes wherever the M0 it was started with pointed).
                // (M0 is the compiler's to use: it is named as clobbered; no instantiation of this kernel uses it)
#define WN_WIPE4(o) "ds_write_addtid_b32 %1 offset:" #o "\n\tds_write_addtid_b32 %1 offset:" #o "+256\n\tds_write_addtid_b32 %1 offset:" #o "+512\n\tds_write_addtid_b32 %1 offset:" #o "+768\n\t"
#define WN_WIPE32(o) WN_WIPE4(o) WN_WIPE4(o + 1024) WN_WIPE4(o + 2048) WN_WIPE4(o + 3072) WN_WIPE4(o + 4096) WN_WIPE4(o + 5120) WN_WIPE4(o + 6144) WN_WIPE4(o + 7168)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t" WN_WIPE32(0) : : "s"(bmbase.base), "v"(vzero) : "m0", "memory");
#undef WN_WIPE32
#undef WN_WIPE4
            }
#endif
            PROF_T(t_5);
            PROF_ADD(7, t_w2, t_5);
            // ---- the shared threshold polled a window ago; then, in this order: the next poll P(w), the word of this window's open
            // pass G(w), the runs of the window after next R(w + 2)
            {
                wn_wait_theta<MT>(pgv);
                const unsigned long long pg = ((unsigned long long)uni((uint32_t)(pgv >> 32)) << 32) | uni((uint32_t)pgv);
                if (pg > th) th = pg;
            }
            wn_load_theta(pgv, vzero, &bt.theta[q]);
            c1_request(pa);
            c1_request(pb);
            load_runs(cur, w - w_lo + 2u);
            PROF_T(t_6);
            PROF_ADD(11, t_5, t_6);
            if (failed) return false;
            return true;
        };
        PROF_T(t_loop);
        for (uint32_t w = w_lo; w < w_hi; w += 2u) {
            if (!window(bufa, w)) break;
            if (w + 1u < w_hi && !window(bufb, w + 1u)) break;
        }
        PROF_T(t_loop_end);
        // (the cold pass's first look -- the upper bounds of every term's first 64 blocks of the item -- is requested here: its round trip
        // to memory runs behind the drain of the loop's last loads and the two open passes' arithmetic)
        double cold_ub[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const uint32_t pS = (uint32_t)__builtin_amdgcn_readlane((int)wS, t), pE = (uint32_t)__builtin_amdgcn_readlane((int)wE, t);
            const uint32_t fbt = (uint32_t)__builtin_amdgcn_readlane((int)fb, t);
            const bool in = (uint32_t)t < mm && pE != pS && (pS >> 7) + lane <= ((pE - 1u) >> 7);
            cold_ub[t] = in ? ix.blk_ub[fbt + (pS >> 7) + lane] : 0.0;
        }
        wn_drain<MT>(bufa, bufb, pa.gw, pb.gw, pgv);
        // (a launch of no more items than waves: nobody draws -- thousands of waves finishing together would queue at the counter)
        uint32_t next_draw = n_items;
        if (lane == 0 && n_items > n_waves) next_draw = atomicAdd(cold_args()->bt.work_ctr, 1u);  // (consumed at the item's very end)
        if (pa.valid && !failed) c2(pa);
        if (pb.valid && !failed) c2(pb);
        failed = failed || __ballot(notf) != 0ull;

        // ---- cold pass (search.rs:203): the item's blocks whose upper bound reaches the threshold of now.  Their postings are scored
        // on their own; one that would enter the list is offered unless its document has a posting in another list of the query --
        // those documents are the completion's (never offered with a partial score).  Everything here is rare: the windows have
        // raised the threshold above most blocks' bounds by now.
        if (!failed && !(dbg & 1u)) {
            // the first 64 blocks of every term at once (one round trip for all the terms, not one each): lane = block
            uint32_t hm0l = 0, hm0h = 0;  // lane = term: the mask of its hot blocks among them
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const uint32_t pS = (uint32_t)__builtin_amdgcn_readlane((int)wS, t), pE = (uint32_t)__builtin_amdgcn_readlane((int)wE, t);
                const uint32_t fbt = (uint32_t)__builtin_amdgcn_readlane((int)fb, t);
                const bool in = (uint32_t)t < mm && pE != pS && (pS >> 7) + lane <= ((pE - 1u) >> 7);
                const double ub = cold_ub[t];  // (requested before the drain)
                const unsigned long long hm = __ballot(in && (unsigned long long)__double_as_longlong(ub) >= th);
                if (lane == (uint32_t)t) {
                    hm0l = (uint32_t)hm;
                    hm0h = (uint32_t)(hm >> 32);
                }
            }
            for (uint32_t t = 0; t < mm; ++t) {
                const uint32_t pS = (uint32_t)__builtin_amdgcn_readlane((int)wS, (int)t), pE = (uint32_t)__builtin_amdgcn_readlane((int)wE, (int)t);
                if (pE == pS) continue;
                const uint32_t fbt = (uint32_t)__builtin_amdgcn_readlane((int)fb, (int)t);
                const double s0t = readlane_f64(s0, t);
                const uint32_t bS = pS >> 7, bE = (pE - 1u) >> 7;  // the item's blocks of the term: bS .. bE
                for (uint32_t b0 = bS; b0 <= bE; b0 += 64u) {
                    unsigned long long hm = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hm0h, (int)t) << 32 |
                                            (uint32_t)__builtin_amdgcn_readlane((int)hm0l, (int)t);
                    if (b0 != bS) {  // (an item of more than 64 blocks of the term)
                        const bool in = b0 + lane <= bE;
                        double ub = 0.0;
                        if (in) ub = ix.blk_ub[fbt + b0 + lane];
                        hm = __ballot(in && (unsigned long long)__double_as_longlong(ub) >= th);
                    }
                    for (; hm != 0ull; hm &= hm - 1ull) {
                        const uint32_t blk = fbt + b0 + (uint32_t)__ffsll((long long)hm) - 1u;
                        const unsigned long long ubb = (unsigned long long)__double_as_longlong(ix.blk_ub[blk]);
                        if ((((unsigned long long)uni((uint32_t)(ubb >> 32)) << 32) | uni((uint32_t)ubb)) < th) continue;  // (the threshold of now)
                        const uint4 bm = uni4(ix.blk_meta[blk]);
                        const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
                        const uint8_t *body = ix.blob + 8ull * bm.z;
                        uint32_t dd0, dd1, tt0, tt1;
                        decode_doc_ids(body, md, n, bm.x, lane, dd0, dd1);
                        decode_fields(body + ((payload_bytes(md, n) + 7u) & ~7u), mt, n, lane, tt0, tt1);
                        const uchar2 fnn = reinterpret_cast<const uchar2 *>(ix.post_fn + 128ull * blk)[lane];
#pragma nounroll
                        for (uint32_t j = 0; j < 2u; ++j) {
                            const uint32_t d = j ? dd1 : dd0, wd = d >> 16;
                            const double tf = (double)(2u * lane + j < n ? (j ? tt1 : tt0) : 1u);
                            const double sc = (tf * s0t) / (tf + S1[j ? fnn.y : fnn.x]);
                            // a posting of the block inside the item's windows that would enter the list
                            bool cand = 2u * lane + j < n && wd >= w_lo && wd < w_hi && (unsigned long long)__double_as_longlong(sc) >= th &&
                                        (rtop.cnt < k || better(sc, d, rtop.kth_s, rtop.kth_d));
                            if (!__ballot(cand)) continue;
                            for (uint32_t t2 = 0; t2 < mm; ++t2) {  // ... unless another list of the query holds the document
                                if (t2 == t) continue;
                                const uint32_t wb2 = (uint32_t)__builtin_amdgcn_readlane((int)wbs, (int)t2);
                                const uint16_t *run = ids16 + 128ull * (uint32_t)__builtin_amdgcn_readlane((int)fbi, (int)t2);
                                uint32_t lo = 0, hi = 0;
                                if (cand) {
                                    lo = ix.win_off[wb2 + wd];
                                    hi = ix.win_off[wb2 + wd + 1u];
                                }
                                while (__ballot(lo < hi)) {  // first posting of the window's run with id >= the document's
                                    const uint32_t mid = (lo + hi) >> 1;
                                    if (lo < hi) {
                                        if ((uint32_t)run[mid] < (d & 0xffffu)) lo = mid + 1u;
                                        else hi = mid;
                                    }
                                }
                                if (cand && lo < ix.win_off[wb2 + wd + 1u] && (uint32_t)run[lo] == (d & 0xffffu)) cand = false;
                            }
                            offer(cand, sc, d);
                        }
                    }
                }
            }
        }
        if (rtop.cnt >= k) {  // the item's k-th score to the query's shared threshold
            const unsigned long long kb = (unsigned long long)__double_as_longlong(rtop.kth_s);
            if (kb > published && lane == 0) atomicMax(&bt.theta[q], kb);
        }

#ifdef VBM25_PROFILE
        {
            const unsigned long long t_end = __builtin_readcyclecounter();
            prof[12] += 1;
            prof[9] += t_loop_end - t_loop;
            prof[8] += t_loop - t_item;
            prof[13] += t_end - t_loop_end;
        }
#endif
        // ---- item result: one list
        const uint32_t nres = failed ? 0u : rtop.cnt;
        const KernArgsP ce = cold_args();
        const size_t list = (size_t)item * ce->bt.lpi;
        // (the list is read by another wave of this launch when the kernel merges: its stores go through to where every XCD sees them
        // -- agent-scope stores -- and are waited for before the query's counter moves.  A release FENCE would write the whole L2 back,
        // once per item: measured, 0.19 -> 0.28 ms on C3.)
#pragma unroll
        for (int r = 0; r < RK; ++r)
            if (r * 64 + lane < nres) {
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(&ce->bt.res_score[list * k + r * 64 + lane]),
                                   (unsigned long long)__double_as_longlong(rtop.score[r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ce->bt.res_doc[list * k + r * 64 + lane], rtop.doc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        if (lane == 0) {
            __hip_atomic_store(&ce->bt.res_cnt[list], nres, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ce->bt.item_failed[item], failed ? 0x101u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (failed) *ce->bt.fail_any = 1u;
        }
        if (ce->bt.win_fuse) {
            // the wave that finishes the query's last item merges the query's lists and writes its records (a counter per query finds
            // it) -- BEHIND the item loop: the host asks for the in-kernel merge only when no wave gets a second item
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t done = 0;
            if (lane == 0) done = atomicAdd(&ce->bt.fused_state[1 + q], 1u);
            if (uni(done) == g - 1u) merge_q = q;
        }
        drawn = ((dbg & 8u) ? 0u : n_waves) + uni(next_draw);
    }
    if (merge_q != NONE32)
        wn_merge_query<RK>(ix, merge_q, g, k, lane, bm, reinterpret_cast<double *>(&S.stage[0]), reinterpret_cast<uint32_t *>(&S.stage[256]));
#ifdef VBM25_PROFILE
    if (bt.prof && lane == 0) {
        unsigned long long *o = bt.prof + ((size_t)blockIdx.x * WN_WAVES + (threadIdx.x >> 6)) * 16;
        for (int i = 0; i < 16; ++i) o[i] = prof[i];
        o[15] = __builtin_readcyclecounter() - prof_t0;
    }
#endif
}
#pragma clang diagnostic pop
