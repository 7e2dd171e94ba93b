// scan_many.h -- scan_many_kernel: more than 16 terms (up to MAX_TERMS), every query of 256 < k <= 1024, dense queries when
// scan_dense_kernel is off, and the items the first-choice kernels gave up.  Exhaustive: term-phased exact f64 accumulation
// in dense doc windows or an LDS hash, LDS top-k, the query's shared threshold.
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25, in
// this order: device_types, decode, plan, topk_lds, block_fetch, topk_reg, scan_range, scan_dense, scan_many, merge.

// ---------------------------------------------------------------------------
// Posting scan
// ---------------------------------------------------------------------------
template <int KMAX>
__global__ void __launch_bounds__(WG) scan_many_kernel(DevIndex ix, DevBatch bt) {
    __shared__ uint32_t s_key[SLOTS];
    __shared__ double s_val[SLOTS];
    __shared__ uint16_t s_cand[SLOTS];
    __shared__ double s_s1[256];
    __shared__ TopK<KMAX> s_top;
    __shared__ uint32_t t_cur[MAX_TERMS], t_end[MAX_TERMS], t_quota[MAX_TERMS];
    __shared__ double t_s0[MAX_TERMS];
    __shared__ uint32_t s_m, s_hi, s_next_lo, s_cand_cnt, s_dense;
    __shared__ unsigned long long s_theta, s_sumdf;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t k = bt.k;
    for (int i = tid; i < 256; i += WG) s_s1[i] = ix.s1[i];

    if (!bt.many_expected && __hip_atomic_load(bt.fail_any, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // (a small grid then)
    const uint32_t n_items = *bt.n_items;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const Item it = bt.items[item];
        const uint32_t others_max = bt.range_max_terms;  // up to that many terms: the other kernels' (0: everything is this kernel's)
        const uint32_t mt_item = bt.dense_on ? (it.m & ~ITEM_DENSE) : it.m;  // dense items: scan_dense_kernel's
        const bool failed = mt_item <= others_max && bt.item_failed[item] != 0;
        if (mt_item <= others_max && !failed) continue;
        const bool force_dense = failed || (it.m & ITEM_DENSE) != 0;
        const uint32_t q = it.q, clo = it.doc_lo, chi = it.doc_hi;
        __syncthreads();  // previous item fully done with LDS
        if (tid == 0) {
            // valid terms of the query, ascending (Query::new guarantees sorted keys)
            uint32_t m = 0;
            unsigned long long sum = 0;
            for (uint32_t p = bt.q_off[q]; p < bt.q_off[q + 1]; ++p) {
                const uint32_t term = bt.term_ids[p];
                if (term >= ix.n_terms) continue;  // search.rs:59-61
                if (m < MAX_TERMS) {
                    t_cur[m] = term;  // resolved below
                    sum += ix.term_df[term];
                    ++m;
                }
            }
            s_m = m;
            s_sumdf = sum;
            s_top.count = 0;
            s_dense = (force_dense || m >= (uint32_t)CAP_BLOCKS) ? 1u : 0u;
        }
        __syncthreads();
        const uint32_t m = s_m;
        const bool dense = s_dense != 0;
        for (uint32_t tt = tid; tt < m; tt += WG) {  // (up to MAX_TERMS terms: four per thread)
            const uint32_t term = t_cur[tt];
            const uint32_t b0 = ix.term_first_block[term], b1 = ix.term_first_block[term + 1];
            // first block whose max_doc >= clo
            uint32_t lo = b0, hi = b1;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ix.blk_max_doc[mid] < clo) lo = mid + 1; else hi = mid;
            }
            t_end[tt] = b1;
            t_s0[tt] = ix.term_s0[term];
            const unsigned long long df = ix.term_df[term];
            const uint32_t share = (uint32_t)(((unsigned long long)(CAP_BLOCKS - (dense ? 0 : (int)m)) * df) / s_sumdf);
            t_quota[tt] = share > 1 ? share : 1;
            t_cur[tt] = lo;
        }
        __syncthreads();

        uint32_t lo = clo;
        unsigned long long published = 0;
        while (lo < chi) {
            // ---- tile bounds + table reset
            if (tid == 0) {
                s_hi = dense ? (chi - lo > (uint32_t)SLOTS ? lo + SLOTS : chi) : chi;
                s_cand_cnt = 0;
                s_next_lo = chi;
                s_theta = __hip_atomic_load(&bt.theta[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int i = tid; i < SLOTS; i += WG) s_key[i] = EMPTY;
            __syncthreads();
            if (!dense)
                for (uint32_t tt = tid; tt < m; tt += WG) {
                    const uint32_t j = t_cur[tt] + t_quota[tt];
                    if (j < t_end[tt]) atomicMin(&s_hi, ix.blk_min_doc[j]);
                }
            __syncthreads();
            const uint32_t hi = s_hi;

            // ---- accumulate, one term per phase (ascending key order)
            for (uint32_t t = 0; t < m; ++t) {
                const uint32_t jend = t_end[t];
                const double s0 = t_s0[t];
                for (uint32_t j = t_cur[t] + wave; j < jend; j += NW) {
                    const uint4 bm = ix.blk_meta[j];
                    if (bm.x >= hi) break;
                    const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
                    const uint8_t *body = ix.blob + 8ull * bm.z;
                    uint32_t d0, d1, f0, f1;
                    const uint8_t *tbody = body + ((payload_bytes(md, n) + 7u) & ~7u);
                    if (md < 32u && mt < 32u) {  // a full bit-packed block: paired 8-byte fetches, DPP prefix sum
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
                    const uchar2 fn = reinterpret_cast<const uchar2 *>(ix.post_fn + 128ull * j)[lane];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const uint32_t i = 2 * lane + e;
                        const uint32_t d = e ? d1 : d0;
                        const uint32_t tfv = e ? f1 : f0;
                        const uint32_t f = e ? fn.y : fn.x;
                        if (i < n && d >= lo && d < hi) {
                            const double tf = (double)tfv;
                            const double p = (tf * s0) / (tf + s_s1[f]);  // bm25.rs:355-358
                            const uint32_t key = d - lo;
                            if (dense) {
                                // direct index: within a term's phase a document has one posting, between phases
                                // there is a barrier -- no atomics needed
                                if (s_key[key] == EMPTY) {
                                    s_key[key] = key;
                                    s_val[key] = p;
                                } else {
                                    s_val[key] += p;
                                }
                                continue;
                            }
                            uint32_t slot = (key * 0x9E3779B1u) >> (32 - SLOTS_LOG2);
                            for (;;) {
                                const uint32_t prev = atomicCAS(&s_key[slot], EMPTY, key);
                                if (prev == EMPTY) {
                                    s_val[slot] = p;
                                    break;
                                }
                                if (prev == key) {
                                    s_val[slot] += p;
                                    break;
                                }
                                slot = (slot + 1) & (SLOTS - 1);
                            }
                        }
                    }
                }
                __syncthreads();
            }

            // ---- candidates of this tile
            {
                const unsigned long long theta = s_theta;
                const uint32_t n = s_top.count;
                const double ws = n >= k ? s_top.score[k - 1] : 0.0;
                const uint32_t wd = n >= k ? s_top.doc[k - 1] : 0u;
                for (int i = tid; i < SLOTS; i += WG) {
                    const uint32_t key = s_key[i];
                    if (key == EMPTY) continue;
                    const double sc = s_val[i];
                    if ((unsigned long long)__double_as_longlong(sc) < theta) continue;
                    if (n >= k && !better(sc, lo + key, ws, wd)) continue;
                    const uint32_t at = atomicAdd(&s_cand_cnt, 1u);
                    s_cand[at] = (uint16_t)i;
                }
            }
            __syncthreads();
            if (wave == 0) {
                const uint32_t cnt = s_cand_cnt;
                for (uint32_t base = 0; base < cnt; base += 64) {
                    const bool has = base + lane < cnt;
                    double sc = 0;
                    uint32_t d = 0;
                    if (has) {
                        const uint32_t slot = s_cand[base + lane];
                        sc = s_val[slot];
                        d = lo + s_key[slot];
                    }
                    topk_offer<KMAX>(s_top, k, has, sc, d, lane);
                }
                if (s_top.count >= k && lane == 0) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(s_top.score[k - 1]);
                    if (bits > published) {
                        atomicMax(&bt.theta[q], bits);
                        published = bits;
                    }
                }
            }
            // ---- advance cursors; next tile starts at the first remaining posting
            for (uint32_t tt = tid; tt < m; tt += WG) {
                uint32_t j = t_cur[tt];
                const uint32_t e = t_end[tt];
                // first block whose max_doc >= hi: gallop, then bisect (a dense term has tens of blocks per
                // window; walking them one dependent load at a time was a fifth of the window's time)
                if (j < e && ix.blk_max_doc[j] < hi) {
                    uint32_t lo_b = j + 1, hi_b = e;
                    for (uint32_t step = 2; lo_b < hi_b; step *= 2) {
                        const uint32_t p = min(lo_b + step - 1, hi_b - 1);
                        if (ix.blk_max_doc[p] < hi) lo_b = p + 1;
                        else {
                            hi_b = p;
                            break;
                        }
                    }
                    while (lo_b < hi_b) {
                        const uint32_t mid = (lo_b + hi_b) >> 1;
                        if (ix.blk_max_doc[mid] < hi) lo_b = mid + 1; else hi_b = mid;
                    }
                    j = lo_b;
                }
                t_cur[tt] = j;
                if (j < e) atomicMin(&s_next_lo, max(hi, ix.blk_min_doc[j]));
            }
            __syncthreads();
            lo = max(hi, s_next_lo);
        }

        // ---- chunk result
        __syncthreads();
        {
            const uint32_t n = s_top.count;
            for (uint32_t i = tid; i < n; i += WG) {
                bt.res_score[(size_t)item * bt.lpi * k + i] = s_top.score[i];
                bt.res_doc[(size_t)item * bt.lpi * k + i] = s_top.doc[i];
            }
            if (tid == 0) bt.res_cnt[(size_t)item * bt.lpi] = n;
        }
    }
}
