// plan.h -- post_fn_kernel (index preparation + validation) and plan_kernel (work items).
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, block_fetch, topk_reg, scan_range, scan_dense, scan_many, merge.

// ---------------------------------------------------------------------------
// Index preparation: fieldnorm of every posting + structural validation
// ---------------------------------------------------------------------------
struct PostFnArgs {
    uint32_t n_blocks, n_docs, n_terms;
    const uint4 *blk_meta;
    const uint8_t *blob, *doc_fieldnorm;
    uint8_t *post_fn;
    uint32_t *post_rel16, *post_tfn;
    uint32_t *post_id16, *win_off;   // scan_win_kernel's planes (device_types.h); term_win says which terms have a table
    const uint32_t *term_win;
    uint32_t n_win;
    uint32_t *error_flag;
    // upper bounds to verify: the scan kernels prune with them
    const uint32_t *term_first_block, *term_wand_tf;
    const uint8_t *term_wand_fn;
    const double *term_s0, *s1, *blk_ub;
    const double *blk_raw;  // the block maxima without the margin (NULL: no block WAND pairs): what "attained" is tested against
};
__global__ void __launch_bounds__(256) post_fn_kernel(PostFnArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t j = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (j >= a.n_blocks) return;
    const uint4 m = a.blk_meta[j];
    const uint32_t n = m.w & 0xff, md = (m.w >> 8) & 0xff, mt = (m.w >> 16) & 0xff;
    const uint8_t *body = a.blob + 8ull * m.z;
    uint32_t d0, d1;
    decode_doc_ids(body, md, n, m.x, lane, d0, d1);
    const uint32_t i0 = 2 * lane, i1 = i0 + 1;
    uint8_t f0 = 0, f1 = 0;
    bool bad = false;
    if (i0 < n) {
        bad |= d0 >= a.n_docs;
        if (d0 < a.n_docs) f0 = a.doc_fieldnorm[d0];
    }
    if (i1 < n) {
        bad |= d1 >= a.n_docs || d1 <= d0;
        if (d1 < a.n_docs) f1 = a.doc_fieldnorm[d1];
    }
    // strictly increasing across lanes, first = min_doc, last = max_doc
    const uint32_t prev = __shfl_up(d1, 1);
    if (lane > 0 && i0 < n) bad |= d0 <= prev;
    if (i0 == 0) bad |= d0 != m.x;
    if (i0 == n - 1) bad |= d0 != m.y;
    if (i1 == n - 1) bad |= d1 != m.y;
    if (bad) atomicOr(a.error_flag, 1u);
    reinterpret_cast<uchar2 *>(a.post_fn + 128ull * j)[lane] = make_uchar2(f0, f1);
    if (a.post_rel16) a.post_rel16[64ull * j + lane] = rel16_block(m.x, m.y, m.w) ? (d1 - m.x) << 16 | (d0 - m.x) : 0u;

    // The WAND pairs must bound every posting: Cache::evaluate of each posting against the block's
    // bound (blk_ub, margin included) and the token's (search.rs:363,377-380).
    uint32_t lo = 0, hi = a.n_terms;  // the term of block j: term_first_block[t] <= j < [t + 1]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.term_first_block[mid] <= j) lo = mid; else hi = mid;
    }
    // The window-major planes (scan_win_kernel): the low 16 bits of every id in posting order, and -- for the terms that have a
    // table -- the number of the term's postings below every multiple of 2^16 documents.  Posting i of block j is posting
    // 128 (j - first block) + i of its term: only a term's last block is short (flag 8 otherwise: the index then goes without
    // these planes).  Every boundary w << 16 lies between two consecutive postings of the term (or before its first / after its
    // last one): the later of the two writes entry w.
    // (an index made without post_id16 keeps the tables: decode_id16_kernel makes the batch's terms' ids per launch)
    if (a.post_id16) a.post_id16[64ull * j + lane] = (d1 & 0xffffu) << 16 | (d0 & 0xffffu);
    if (a.win_off) {
        const uint32_t fb = a.term_first_block[lo], fe = a.term_first_block[lo + 1];
        if (j + 1 < fe && n != 128u) atomicOr(a.error_flag, 8u);
        const uint32_t wb = a.term_win[lo];
        if (wb != NONE32 && !bad) {
            const long long before = j == fb ? -1ll : (long long)a.blk_meta[j - 1].y;  // last document of the term before this block
            if (lane == 0 && before >= (long long)d0) atomicOr(a.error_flag, 1u);  // a term's blocks must ascend
            const long long p0 = lane == 0 ? before : (long long)prev;
            const long long nw = (long long)a.n_win;
            const uint32_t at = 128u * (j - fb);
            if (i0 < n)
                for (long long w = (p0 >> 16) + 1; w <= min((long long)(d0 >> 16), nw); ++w) a.win_off[wb + w] = at + i0;
            if (i1 < n)
                for (long long w = (long long)(d0 >> 16) + 1; w <= min((long long)(d1 >> 16), nw); ++w) a.win_off[wb + w] = at + i1;
            if (j + 1 == fe && lane == (n - 1u) >> 1) {  // behind the term's last posting: all of them
                const uint32_t dl = ((n - 1u) & 1u) ? d1 : d0;
                for (long long w = (long long)(dl >> 16) + 1; w <= nw; ++w) a.win_off[wb + w] = at + n;
            }
        }
    }
    const double s0 = a.term_s0[lo];
    const double wtf = (double)a.term_wand_tf[lo];
    const double tub = ((wtf * s0) / (wtf + a.s1[a.term_wand_fn[lo]])) * (1.0 + 1e-12);
    const double bub = a.blk_ub[j];
    uint32_t t0, t1;
    decode_fields(body + ((payload_bytes(md, n) + 7u) & ~7u), mt, n, lane, t0, t1);
    // (the two term frequencies and fieldnorm bytes of the lane's postings.  scan_range_kernel / scan_dense_kernel read the word of
    // tfn_block blocks only; scan_win_kernel reads it for every posting -- tails included -- and takes a zero term frequency for
    // "wider than a byte")
    a.post_tfn[64ull * j + lane] = t0 <= 255u && t1 <= 255u ? t0 | t1 << 8 | (uint32_t)f0 << 16 | (uint32_t)f1 << 24 : 0u;
    // (attained: the very score the k-th largest maxima are taken from -- blk_raw, no margin: compared after the margin two scores
    // an ulp apart could round to the same value and a bound one ulp above every posting would pass)
    const bool braw_ok = a.blk_raw != nullptr;
    const double braw = braw_ok ? a.blk_raw[j] : 0.0;
    bool loose = false, attained = false;  // attained: the block's bound is the score of one of its postings (flag 4 if not: no error,
                                           // but the k-th largest block maxima are then no lower bound of anything)
    if (i0 < n) {
        const double tf = (double)t0, p = (tf * s0) / (tf + a.s1[f0]);
        loose |= p > bub || p > tub;
        attained |= braw_ok && p == braw;
    }
    if (i1 < n) {
        const double tf = (double)t1, p = (tf * s0) / (tf + a.s1[f1]);
        loose |= p > bub || p > tub;
        attained |= braw_ok && p == braw;
    }
    if (loose) atomicOr(a.error_flag, 2u);
    if (!__ballot(attained) && lane == 0) atomicOr(a.error_flag, 4u);
}

// ---------------------------------------------------------------------------
// Index creation: what the index derives from the raw block / token arrays, on the device (the arrays may never have been on
// the host: vbm25_index_create_from_device)
// ---------------------------------------------------------------------------
struct DeriveArgs {
    uint32_t n_blocks, n_terms, has_wand;
    const uint32_t *term_first_block, *term_wand_tf, *blk_min_doc, *blk_max_doc, *blk_off8, *blk_wand_tf;
    const uint8_t *term_wand_fn, *blk_n, *blk_meta_doc, *blk_meta_tf, *blk_wand_fn;
    const double *term_s0, *s1;
    uint4 *blk_meta;
    double *blk_ub;   // Cache::evaluate(block WAND pair) x (1 + 1e-12) (search.rs:377-380 evaluates it per visited block; here once)
    double *blk_raw;  // the same without the margin (has_wand): what the k-th largest block maxima of a term are taken from
};
__global__ void __launch_bounds__(256) blk_derive_kernel(DeriveArgs a) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.n_blocks) return;
    uint32_t lo = 0, hi = a.n_terms;  // the term of block j: term_first_block[t] <= j < [t + 1]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.term_first_block[mid] <= j) lo = mid; else hi = mid;
    }
    const uint32_t wfn = a.has_wand ? a.blk_wand_fn[j] : 0u;
    a.blk_meta[j] = make_uint4(a.blk_min_doc[j], a.blk_max_doc[j], a.blk_off8[j],
                               (uint32_t)a.blk_n[j] | (uint32_t)a.blk_meta_doc[j] << 8 | (uint32_t)a.blk_meta_tf[j] << 16 | wfn << 24);
    const double s0 = a.term_s0[lo];
    double ub;
    if (a.has_wand) {
        const double tf = (double)a.blk_wand_tf[j];
        ub = (tf * s0) / (tf + a.s1[wfn]);
        a.blk_raw[j] = ub;
    } else {
        const double wtf = (double)a.term_wand_tf[lo];
        ub = (wtf * s0) / (wtf + a.s1[a.term_wand_fn[lo]]);
    }
    a.blk_ub[j] = ub * (1.0 + 1e-12);  // margin: another posting's evaluate may round one ulp higher
}
// sorted: every term's block maxima in descending order (segmented sort) -> the 2^i-th largest, i = 0..8 (0 when fewer)
__global__ void __launch_bounds__(256) kth_pick_kernel(uint32_t n_terms, const uint32_t *term_first_block, const double *sorted, double *kth) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_terms * 9u) return;
    const uint32_t t = e / 9u, i = e - 9u * t;
    const uint32_t b0 = term_first_block[t], nb = term_first_block[t + 1] - b0, at = (1u << i) - 1u;
    kth[e] = at < nb ? sorted[b0 + at] : 0.0;
}
// ---------------------------------------------------------------------------
// Planner
// ---------------------------------------------------------------------------
// Block-wide inclusive scan of one u64 per thread (PLAN_WG threads): wave scans + one LDS hop.
__device__ __forceinline__ unsigned long long plan_incl_scan(unsigned long long v, unsigned long long *s_wave,
                                                             unsigned long long &total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long y = __shfl_up(v, o);
        if ((int)lane >= o) v += y;
    }
    if (lane == 63) s_wave[wave] = v;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    for (uint32_t w = 0; w < PLAN_WG / 64; ++w) {
        const unsigned long long x = s_wave[w];
        if (w < wave) before += x;
        all += x;
    }
    __syncthreads();
    total = all;
    return v + before;
}

// dense_c != 0: every dense query (q_dense) is cut into dense_c items of equal document counts -- the dense-window kernel's
// work per item goes with its windows, not with its postings -- and does not count towards the other queries' chunk size.
__global__ void __launch_bounds__(PLAN_WG) plan_kernel(DevIndex ix, DevBatch bt, uint32_t max_items,
                                                       uint32_t target_items, uint32_t min_chunk, uint32_t dense_c) {
    __shared__ unsigned long long s_wave[PLAN_WG / 64];
    __shared__ uint32_t s_bucket[PLAN_BUCKETS];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < (uint32_t)PLAN_BUCKETS; i += PLAN_WG) s_bucket[i] = 0;
    const uint32_t per = (bt.nq + PLAN_WG - 1) / PLAN_WG;
    const uint32_t q0 = min(bt.nq, tid * per), q1 = min(bt.nq, q0 + per);
    // per-launch state of the scan kernels (saves two memset launches per step)
    for (uint32_t i = tid; i < bt.nq; i += PLAN_WG) bt.theta[i] = 0;
    for (uint32_t i = tid; i < max_items; i += PLAN_WG) {
        bt.item_failed[i] = 0;
        bt.item_order[i] = i;  // (every slot defined even when the item count overflows: error_flag 2)
    }
    __syncthreads();
    if (bt.lpi > 1)  // lists a kernel does not write must read as empty
        for (uint32_t i = tid; i < max_items * bt.lpi; i += PLAN_WG) bt.res_cnt[i] = 0;
    if (tid == 0) {
        bt.work_ctr[0] = 0;
        bt.work_ctr[1] = 0;
        *bt.fail_any = 0;
    }

    auto postings_of = [&](uint32_t q) {
        unsigned long long t = 0;
        for (uint32_t p = bt.q_off[q]; p < bt.q_off[q + 1]; ++p) {
            uint32_t term = bt.term_ids[p];
            if (term < ix.n_terms) t += ix.term_df[term];
        }
        return t;
    };
    // a thread usually owns one query (nq <= PLAN_WG): its posting count is read once and kept -- every
    // evaluation is a chain of three dependent global loads, which is what a small batch's plan costs
    const bool one = q1 - q0 == 1;
    const unsigned long long own_postings = one ? postings_of(q0) : 0ull;
    auto is_dense = [&](uint32_t q) { return dense_c != 0 && bt.q_dense[q] != 0; };
    unsigned long long local = one && !is_dense(q0) ? own_postings : 0ull;
    if (!one)
        for (uint32_t q = q0; q < q1; ++q)
            if (!is_dense(q)) local += postings_of(q);
    unsigned long long total = 0;
    plan_incl_scan(local, s_wave, total);
    unsigned long long chunk = (total + target_items - 1) / target_items;
    if (chunk < min_chunk) chunk = min_chunk;
    auto chunks_of = [&](uint32_t q) -> uint32_t {
        unsigned long long t = one ? own_postings : postings_of(q);
        if (t == 0) return 0u;
        if (is_dense(q)) return min(dense_c, ix.n_docs);
        // nearest, not ceil: a batch of similar queries gets the same count for all of them, i.e. the
        // item count lands on the target (a multiple of the resident waves) instead of ~8 % above it
        unsigned long long c = (t + chunk / 2) / chunk;
        if (c == 0) c = 1;
        if (c > ix.n_docs) c = ix.n_docs;
        return (uint32_t)c;
    };
    unsigned long long cnt = 0;
    for (uint32_t q = q0; q < q1; ++q) cnt += chunks_of(q);
    unsigned long long run = 0;
    const unsigned long long incl = plan_incl_scan(cnt, s_wave, run);
    if (tid == 0) {
        *bt.n_items = (uint32_t)min(run, (unsigned long long)max_items);
        if (run > max_items) atomicOr(bt.error_flag, 2u);
        bt.q_item_base[bt.nq] = (uint32_t)min(run, (unsigned long long)max_items);
    }
    uint32_t base = (uint32_t)(incl - cnt);
    // The persistent scan kernels draw items through item_order: longest first (the last items drawn decide when the
    // launch ends).  A counting sort over buckets of the items' postings: exponent and three mantissa bits.
    auto bucket_of = [&](unsigned long long postings, uint32_t c) -> uint32_t {
        const unsigned long long w = postings / (c ? c : 1u) + 1ull;
        const uint32_t e = 63u - (uint32_t)__builtin_clzll(w);
        const uint32_t sub = e >= 3u ? (uint32_t)(w >> (e - 3u)) & 7u : 0u;
        return min(8u * e + sub, (uint32_t)PLAN_BUCKETS - 1u);
    };
    for (uint32_t q = q0; q < q1; ++q) {
        const uint32_t c = chunks_of(q);
        if (c) atomicAdd(&s_bucket[bucket_of(one ? own_postings : postings_of(q), c)], c);
    }
    __syncthreads();
    {   // first position of every bucket, from the largest bucket down (PLAN_BUCKETS <= PLAN_WG: one bucket per thread)
        const uint32_t bkt = (uint32_t)PLAN_BUCKETS - 1u - tid;
        const unsigned long long v = tid < (uint32_t)PLAN_BUCKETS ? s_bucket[bkt] : 0u;
        unsigned long long all = 0;
        const unsigned long long in = plan_incl_scan(v, s_wave, all);
        if (tid < (uint32_t)PLAN_BUCKETS) s_bucket[bkt] = (uint32_t)(in - v);
    }
    __syncthreads();
    for (uint32_t q = q0; q < q1; ++q) {
        const uint32_t c = chunks_of(q);
        uint32_t nterms = 0;
        for (uint32_t p = bt.q_off[q]; p < bt.q_off[q + 1]; ++p) nterms += bt.term_ids[p] < ix.n_terms;
        if (bt.q_dense[q]) nterms |= ITEM_DENSE;
        bt.q_item_base[q] = min(base, max_items);
        const uint32_t pos = c ? atomicAdd(&s_bucket[bucket_of(one ? own_postings : postings_of(q), c)], c) : 0u;
        for (uint32_t i = 0; i < c && base + i < max_items; ++i) {
            Item it;
            it.q = q;
            it.doc_lo = (uint32_t)((unsigned long long)ix.n_docs * i / c);
            it.doc_hi = (uint32_t)((unsigned long long)ix.n_docs * (i + 1) / c);
            it.m = nterms;
            bt.items[base + i] = it;
            if (pos + i < max_items) bt.item_order[pos + i] = base + i;
        }
        base += c;
    }
}
