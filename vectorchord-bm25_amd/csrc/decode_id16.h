// decode_id16.h -- decode_id16_kernel: the low 16 bits of the document ids of the batch's terms, unpacked from the reference's
// bit-packed blocks (compression.rs:65-92; the d1 delta streams of bitpacking_u32_ordered.rs:191-237) into the batch's scratch plane
// ahead of scan_win_kernel.  The route of an index made WITHOUT the post_id16 plane (tuning id16_plane = 0: 2 bytes per posting less in
// HBM): what index creation would have derived once for every posting is made per launch for the postings the batch's queries name.
// Part of libvbm25's device code: included by search.hip inside namespace vbm25 after device_types.h and decode.h.
//
// One workgroup of four waves per term position of the batch (grid = q_off[nq]); a wave takes every fourth block of the term, four at
// a time: the four blocks' metadata come with ONE vector load (lane i = block i) a group ahead, the eight 8-byte loads that hold the
// lanes' two fields of each block are issued together, then each block is two v_alignbit + mask, a DPP prefix sum over the wave
// (decode_doc_ids' own) and one coalesced 256-byte store: word l of a block = (id[2l + 1] & 0xffff) << 16 | (id[2l] & 0xffff), exactly
// what post_fn_kernel writes into the index's plane (plan.h).  Byte-packed tails and raw blocks take decode_doc_ids itself.
#ifndef VBM25_DI_UN
#define VBM25_DI_UN 4
#define VBM25_DI_WAVES 4
#endif
constexpr int DI_UN = VBM25_DI_UN;  // blocks of a wave in flight together
constexpr int DI_WAVES = VBM25_DI_WAVES;
__global__ void __launch_bounds__(DI_WAVES * 64) decode_id16_kernel(DevIndex ix, const uint32_t *term_ids, const uint32_t *id16_fb, uint32_t *dst) {
    const uint32_t lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
    const uint32_t t = uni(term_ids[blockIdx.x]);
    if (t >= ix.n_terms) return;  // search.rs:59-61: a token the segment does not hold
    const uint32_t b0 = uni(ix.term_first_block[t]), b1 = uni(ix.term_first_block[t + 1]);
    uint32_t *out = dst + 64ull * uni(id16_fb[blockIdx.x]);
    // the lane's two fields of a bit-packed block: values 2 lane and 2 lane + 1 are the streams l = 2 lane mod 4 and l + 1 at the same bit
    const uint32_t l = (2u * lane) & 3u, fidx = lane >> 1;
    auto metas = [&](uint32_t j) -> uint4 {  // lane i: the metadata of block j + DI_WAVES i
        const uint32_t jj = j + (uint32_t)DI_WAVES * min(lane, (uint32_t)DI_UN - 1u);
        return ix.blk_meta[min(jj, b1 - 1u)];
    };
    struct Grp {
        uint32_t mx[DI_UN], mz[DI_UN], mw[DI_UN], sh[DI_UN];
        uint2 lo[DI_UN], hi[DI_UN];
    };
    auto issue = [&](Grp &g, const uint4 ml) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DI_UN; ++i) {
            g.mx[i] = (uint32_t)__builtin_amdgcn_readlane((int)ml.x, i);
            g.mz[i] = (uint32_t)__builtin_amdgcn_readlane((int)ml.z, i);
            g.mw[i] = (uint32_t)__builtin_amdgcn_readlane((int)ml.w, i);
            const uint32_t width = (g.mw[i] >> 8) & 31u;  // (a byte-packed or raw block's loads land somewhere in the block or the blob's slack: unused)
            const uint32_t bit = fidx * width, w = bit >> 5;
            g.sh[i] = bit & 31u;
            const uint32_t *w32 = reinterpret_cast<const uint32_t *>(ix.blob + 8ull * g.mz[i]) + 4u * w + l;
            g.lo[i] = *reinterpret_cast<const uint2 *>(w32);
            g.hi[i] = *reinterpret_cast<const uint2 *>(w32 + 4);
        }
    };
    // A group of bit-packed blocks (all but a term's byte-packed tail and the rare raw block) is ONE straight-line block: a branch
    // with a load behind it makes the compiler wait for every load in flight -- the next group's -- at the join.  (A group that
    // reaches beyond the term's last block holds that block again -- metas clamps -- and writes it again: the same words.)
    auto finish = [&](const Grp &g, const uint32_t j) __attribute__((always_inline)) {
        bool allfast = true;
#pragma unroll
        for (int i = 0; i < DI_UN; ++i) {
            const uint32_t md = (g.mw[i] >> 8) & 0xffu;
            allfast = allfast && (md >> 7) == 0u && md != 32u;
        }
        if (allfast) {
            // the four blocks' prefix sums step by step side by side: a DPP instruction must not follow the write of its source
            // directly (two wait states) -- four independent chains fill each other's
            uint32_t v0[DI_UN], own[DI_UN], x[DI_UN];
#pragma unroll
            for (int i = 0; i < DI_UN; ++i) {
                const uint32_t mask = (1u << ((g.mw[i] >> 8) & 31u)) - 1u;  // (width 0: every delta zero)
                v0[i] = __builtin_amdgcn_alignbit(g.hi[i].x, g.lo[i].x, g.sh[i]) & mask;
                own[i] = v0[i] + (__builtin_amdgcn_alignbit(g.hi[i].y, g.lo[i].y, g.sh[i]) & mask);
                x[i] = own[i];
            }
#define DI_STEP(ctrl, rowmask)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < DI_UN; ++i) x[i] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[i], ctrl, rowmask, 0xf, false);
            DI_STEP(0x111, 0xf)  // row_shr:1
            DI_STEP(0x112, 0xf)  // row_shr:2
            DI_STEP(0x114, 0xf)  // row_shr:4
            DI_STEP(0x118, 0xf)  // row_shr:8
            DI_STEP(0x142, 0xa)  // row_bcast:15
            DI_STEP(0x143, 0xc)  // row_bcast:31
#undef DI_STEP
#pragma unroll
            for (int i = 0; i < DI_UN; ++i) {
                const uint32_t jb = min(j + (uint32_t)(DI_WAVES * i), b1 - 1u);
                const uint32_t d0 = g.mx[i] + (x[i] - own[i]) + v0[i], d1 = g.mx[i] + x[i];
#ifdef VBM25_DI_NOSTORE
                if (d0 == 0xdeadbeefu)
#endif
                out[64ull * (jb - b0) + lane] = (d1 & 0xffffu) << 16 | (d0 & 0xffffu);
            }
        } else {
#pragma nounroll
            for (int i = 0; i < DI_UN; ++i) {
                const uint32_t jb = j + (uint32_t)(DI_WAVES * i);
                if (jb >= b1) break;
                const uint4 m = uni4(ix.blk_meta[jb]);
                uint32_t d0, d1;
                decode_doc_ids(ix.blob + 8ull * m.z, (m.w >> 8) & 0xffu, m.w & 0xffu, m.x, lane, d0, d1);
                out[64ull * (jb - b0) + lane] = (d1 & 0xffffu) << 16 | (d0 & 0xffffu);
            }
        }
    };
    // group n + 1's fields are requested before group n is unpacked, the metadata of group n + 2 before that: a wave has two round
    // trips to memory in flight behind the one it works on
    constexpr uint32_t STEP = (uint32_t)(DI_WAVES * DI_UN);
    uint32_t j = b0 + wave;
    if (j >= b1) return;
    Grp g0, g1;
    uint4 mn = metas(j);
    issue(g0, mn);
    mn = metas(j + STEP);
    for (;;) {
        const bool more1 = j + STEP < b1;
        if (more1) {  // (the metadata first: the wait for it next time round leaves this group's eight loads in flight)
            const uint4 mc = mn;
            mn = metas(j + 2u * STEP);
            issue(g1, mc);
        }
        finish(g0, j);
        if (!more1) break;
        j += STEP;
        const bool more0 = j + STEP < b1;
        if (more0) {
            const uint4 mc = mn;
            mn = metas(j + 2u * STEP);
            issue(g0, mc);
        }
        finish(g1, j);
        if (!more0) break;
        j += STEP;
    }
}
