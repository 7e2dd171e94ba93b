// decode_id16.h -- decode_id16_kernel: the low 16 bits of the document ids of the batch's terms, unpacked from the reference's
// bit-packed blocks (compression.rs:65-92; the d1 delta streams of bitpacking_u32_ordered.rs:191-237) into the batch's scratch plane
// ahead of scan_win_kernel.  The route of an index made WITHOUT the post_id16 plane (tuning id16_plane = 0: 2 bytes per posting less in
// HBM): what index creation would have derived once for every posting is made per launch for the postings the batch's queries name.
// Part of libvbm25's device code: included by search.hip inside namespace vbm25 after device_types.h, decode.h and topk_reg.h.
//
// One workgroup of four waves per term position of the batch (grid = q_off[nq]).  A wave takes the term's blocks TWO at a time: lanes
// 0 .. 31 one block, lanes 32 .. 63 the next, FOUR postings per lane -- the four streams of the 4-lane vertical layout at one bit
// position (bitpacking.rs:58-98) are ONE aligned 16-byte row, so a lane's four fields are two 16-byte loads (the row and the one its
// fields may straddle into), four v_alignbit + mask, three adds, a five-step DPP prefix sum over the 32 lanes of its block (the rows
// of 16 lanes and one row broadcast: the 64-lane scan without its last step) and ONE 8-byte store: 32 lanes x 8 bytes = the block's 256
// bytes, word l = (id[2l + 1] & 0xffff) << 16 | (id[2l] & 0xffff) -- exactly what post_fn_kernel writes into the index's plane (plan.h).
// Every lane loads its block's 16 bytes of metadata itself (32 lanes, one address: one request), two pairs ahead; the fields' rows
// one pair ahead.  (Round 6's first version -- one block per 64 lanes, two postings per lane, the metadata through v_readlane: 45
// instructions per block, 0.195 ms on C3; this one: tools/dec_variants_run.sh.)  Byte-packed tails and raw blocks take decode_doc_ids.
constexpr int DI_WAVES = 4;
__global__ void __launch_bounds__(DI_WAVES * 64) decode_id16_kernel(DevIndex ix, const uint32_t *term_ids, const uint32_t *id16_fb, uint32_t *dst) {
    const uint32_t lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
    const uint32_t t = uni(term_ids[blockIdx.x]);
    if (t >= ix.n_terms) return;  // search.rs:59-61: a token the segment does not hold
    const uint32_t b0 = uni(ix.term_first_block[t]), b1 = uni(ix.term_first_block[t + 1]);
    uint32_t *out = dst + 64ull * uni(id16_fb[blockIdx.x]);
    const uint32_t half = lane >> 5, L = lane & 31u;
    // (a pair that reaches beyond the term's last block holds that block again -- the clamp -- and writes it again: the same words)
    auto block_of = [&](uint32_t pair) -> uint32_t { return min(b0 + 2u * pair + half, b1 - 1u); };
    struct Rows {
        uint4 lo, hi;
    };
    auto issue = [&](const uint4 m) -> Rows {
        const uint32_t width = (m.w >> 8) & 31u;  // (a byte-packed or raw block's loads land somewhere in the block or the blob's slack: unused)
        const uint32_t w = (L * width) >> 5;
        const uint4 *row = reinterpret_cast<const uint4 *>(ix.blob + 8ull * m.z) + w;
        Rows r;
        r.lo = row[0];
        r.hi = row[1];
        return r;
    };
    const uint32_t n_pairs = (b1 - b0 + 1u) >> 1;
    constexpr uint32_t STEP = (uint32_t)DI_WAVES;
    uint32_t pair = wave;
    if (pair >= n_pairs) return;
    uint4 m_cur = ix.blk_meta[block_of(pair)];
    Rows r_cur = issue(m_cur);
    uint4 m_next = ix.blk_meta[block_of(pair + STEP)];
    for (; pair < n_pairs; pair += STEP) {
        // the next pair's rows and the metadata of the one after it are requested before this pair is unpacked
        const uint4 m = m_cur;
        const Rows r = r_cur;
        const bool more = pair + STEP < n_pairs;
        if (more) {
            m_cur = m_next;
            m_next = ix.blk_meta[block_of(pair + 2u * STEP)];
            r_cur = issue(m_cur);
        }
        const uint32_t blk = block_of(pair), md = (m.w >> 8) & 0xffu;
        const bool fast = (md >> 7) == 0u && md != 32u;
        if (!__ballot(!fast)) {  // both blocks bit-packed (all but a term's byte-packed tail and the rare raw block): one straight-line block
            const uint32_t sh = (L * md) & 31u, mask = (1u << (md & 31u)) - 1u;  // (width 0: every delta zero)
            const uint32_t v0 = __builtin_amdgcn_alignbit(r.hi.x, r.lo.x, sh) & mask, v1 = __builtin_amdgcn_alignbit(r.hi.y, r.lo.y, sh) & mask;
            const uint32_t v2 = __builtin_amdgcn_alignbit(r.hi.z, r.lo.z, sh) & mask, v3 = __builtin_amdgcn_alignbit(r.hi.w, r.lo.w, sh) & mask;
            const uint32_t own = (v0 + v1) + (v2 + v3);
            uint32_t x = own;  // inclusive prefix sum over the 32 lanes of the lane's block
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
            x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 (rows 1 and 3: the second halves of the two blocks)
            const uint32_t d0 = m.x + (x - own) + v0, d1 = d0 + v1, d2 = d1 + v2, d3 = d2 + v3;
#ifdef VBM25_DI_NOSTORE  // (timing experiment: what the stores cost)
            if (d0 == 0xdeadbeefu)
#endif
            reinterpret_cast<uint2 *>(out + 64ull * (blk - b0))[L] = make_uint2((d1 & 0xffffu) << 16 | (d0 & 0xffffu), (d3 & 0xffffu) << 16 | (d2 & 0xffffu));
        } else {
#pragma nounroll
            for (uint32_t h = 0; h < 2u; ++h) {  // the pair's blocks one after the other, 64 lanes each
                const uint32_t jb = min(b0 + 2u * pair + h, b1 - 1u);
                const uint4 mm = uni4(ix.blk_meta[jb]);
                uint32_t d0, d1;
                decode_doc_ids(ix.blob + 8ull * mm.z, (mm.w >> 8) & 0xffu, mm.w & 0xffu, mm.x, lane, d0, d1);
                out[64ull * (jb - b0) + lane] = (d1 & 0xffffu) << 16 | (d0 & 0xffffu);
            }
        }
    }
}
