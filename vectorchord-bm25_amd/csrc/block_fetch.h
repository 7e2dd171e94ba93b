// block_fetch.h -- split decode: raw words fetched early, fields extracted later; DPP wave helpers.
// Part of libvbm25's single device translation unit: included by search.hip inside namespace vbm25.

// ---------------------------------------------------------------------------
// Split decode used by scan_kernel: the two dwords that hold a field are fetched early
// (possibly one tile ahead) and the field is extracted later.  One formula covers every
// codec of compression.rs:65-136: bit-packed (lane stream words 16 bytes apart), width 32 /
// bytewidth 4 (raw), byte-packed tails (unaligned little-endian bytes).
// ---------------------------------------------------------------------------
struct FieldAddr {
    uint32_t off0, off1, sh, mask;
};
__device__ __forceinline__ FieldAddr field_addr(uint32_t meta, uint32_t n, uint32_t i) {
    FieldAddr a;
    const uint32_t width = meta & 127u;
    if ((meta >> 7) == 0) {
        if (width == 32) {
            a.off0 = a.off1 = 4 * i;
            a.sh = 0;
            a.mask = 0xffffffffu;
        } else {
            const uint32_t l = i & 3, bit = (i >> 2) * width, w = bit >> 5;
            a.sh = bit & 31;
            a.off0 = 16 * w + 4 * l;
            a.off1 = a.off0 + ((a.sh + width > 32) ? 16u : 0u);
            a.mask = (1u << width) - 1u;  // width 0 -> mask 0 -> field 0
        }
    } else {
        const uint32_t bo = (i < n ? i : 0u) * width;
        a.off0 = bo & ~3u;
        a.off1 = a.off0 + 4;
        a.sh = 8 * (bo & 3u);
        a.mask = width >= 4 ? 0xffffffffu : (1u << (8 * width)) - 1u;
    }
    return a;
}
__device__ __forceinline__ uint32_t field_val(uint32_t lo, uint32_t hi, const FieldAddr &a) {
    return __builtin_amdgcn_alignbit(hi, lo, a.sh) & a.mask;  // ((hi:lo) >> sh), sh < 32
}
struct BlockFetch {  // raw dwords of one block for this lane: doc fields 0/1, tf fields 0/1
    uint32_t dlo0, dhi0, dlo1, dhi1, tlo0, thi0, tlo1, thi1;
    uint32_t fn;  // two fieldnorm bytes
};
// Bit-packed blocks: a lane's two values (indices 2L, 2L+1) sit in adjacent lane streams at the
// same step, so their words are one aligned 8-byte pair in group w and one in group w+1.
__device__ __forceinline__ void pair_fetch(const uint8_t *__restrict__ p, uint32_t width, uint32_t lane,
                                           uint32_t &lo0, uint32_t &hi0, uint32_t &lo1, uint32_t &hi1) {
    const uint32_t bit = __umul24(lane >> 1, width);   // step t = (2L) >> 2
    const uint32_t off = 16 * (bit >> 5) + 8 * (lane & 1);  // streams l0 = 2*(L&1), l0 + 1
    const uint2 a = *reinterpret_cast<const uint2 *>(p + off);
    const uint2 b = *reinterpret_cast<const uint2 *>(p + off + 16);  // may be the next payload: unused then
    lo0 = a.x;
    lo1 = a.y;
    hi0 = b.x;
    hi1 = b.y;
}
__device__ __forceinline__ void pair_extract(uint32_t width, uint32_t lane, uint32_t lo0, uint32_t hi0,
                                             uint32_t lo1, uint32_t hi1, uint32_t &v0, uint32_t &v1) {
    const uint32_t sh = __umul24(lane >> 1, width) & 31;
    const uint32_t mask = width >= 32 ? 0xffffffffu : (1u << width) - 1u;
    v0 = __builtin_amdgcn_alignbit(hi0, lo0, sh) & mask;  // ((hi:lo) >> sh), sh < 32
    v1 = __builtin_amdgcn_alignbit(hi1, lo1, sh) & mask;
}
__device__ __forceinline__ void block_fetch(const DevIndex &ix, const uint4 bm, uint32_t j,
                                            uint32_t lane, BlockFetch &f) {
    const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
    const uint8_t *body = ix.blob + 8ull * bm.z;
    const uint8_t *tbody = body + ((payload_bytes(md, n) + 7u) & ~7u);
    if ((md >> 7) == 0) {  // full block (both streams bit-packed, compression.rs:42-52,99-103)
        pair_fetch(body, md & 127u, lane, f.dlo0, f.dhi0, f.dlo1, f.dhi1);
        pair_fetch(tbody, mt & 127u, lane, f.tlo0, f.thi0, f.tlo1, f.thi1);
    } else {               // tail block: byte-packed, generic addressing
        const FieldAddr a0 = field_addr(md, n, 2 * lane), a1 = field_addr(md, n, 2 * lane + 1);
        const FieldAddr b0 = field_addr(mt, n, 2 * lane), b1 = field_addr(mt, n, 2 * lane + 1);
        f.dlo0 = *reinterpret_cast<const uint32_t *>(body + a0.off0);
        f.dhi0 = *reinterpret_cast<const uint32_t *>(body + a0.off1);
        f.dlo1 = *reinterpret_cast<const uint32_t *>(body + a1.off0);
        f.dhi1 = *reinterpret_cast<const uint32_t *>(body + a1.off1);
        f.tlo0 = *reinterpret_cast<const uint32_t *>(tbody + b0.off0);
        f.thi0 = *reinterpret_cast<const uint32_t *>(tbody + b0.off1);
        f.tlo1 = *reinterpret_cast<const uint32_t *>(tbody + b1.off0);
        f.thi1 = *reinterpret_cast<const uint32_t *>(tbody + b1.off1);
    }
    f.fn = reinterpret_cast<const uint16_t *>(ix.post_fn + 128ull * j)[lane];
}
// fields of a fetched block: document-id deltas (or raw ids) and term frequencies
__device__ __forceinline__ void block_fields(const uint4 bm, uint32_t lane, const BlockFetch &f,
                                             uint32_t &v0, uint32_t &v1, uint32_t &f0, uint32_t &f1) {
    const uint32_t n = bm.w & 0xff, md = (bm.w >> 8) & 0xff, mt = (bm.w >> 16) & 0xff;
    if ((md >> 7) == 0) {
        pair_extract(md & 127u, lane, f.dlo0, f.dhi0, f.dlo1, f.dhi1, v0, v1);
        pair_extract(mt & 127u, lane, f.tlo0, f.thi0, f.tlo1, f.thi1, f0, f1);
    } else {
        v0 = field_val(f.dlo0, f.dhi0, field_addr(md, n, 2 * lane));
        v1 = field_val(f.dlo1, f.dhi1, field_addr(md, n, 2 * lane + 1));
        f0 = field_val(f.tlo0, f.thi0, field_addr(mt, n, 2 * lane));
        f1 = field_val(f.tlo1, f.thi1, field_addr(mt, n, 2 * lane + 1));
    }
}

// Reductions over lanes 0..15 (one DPP row); result valid in lane 15, broadcast with readlane.
__device__ __forceinline__ uint32_t row16_min_bcast(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ uint32_t row16_incl_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}

__device__ __forceinline__ double readlane_f64(double v, uint32_t src_lane) {  // src_lane uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), (int)src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), (int)src_lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t wave_shr1_u32(uint32_t v) {  // lane l gets lane l-1 (lane 0: itself)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
__device__ __forceinline__ double wave_shr1_f64(double v) {
    const uint32_t lo = wave_shr1_u32((uint32_t)__double2loint(v));
    const uint32_t hi = wave_shr1_u32((uint32_t)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}

// Inclusive prefix sum / minimum over the wave: DPP row shifts inside 16-lane rows, then row broadcasts across rows.
// All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return x;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {  // uniform result
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x111, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x112, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x114, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x118, 0xf, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x142, 0xa, 0xf, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
