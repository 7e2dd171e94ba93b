// blake3.cpp -- BLAKE3 hash / keyed hash, written from the public specification (the BLAKE3 paper, section 2:
// 7-round compression on 16 words, 1 KiB chunks, binary tree of chaining values).  Needed for
// vector.rs:19-35: lexemes of >= 16 bytes (or containing NUL) are keyed by the first 16 bytes of
// blake3::keyed_hash(seed, lexeme) (crate blake3 1.8.4, crates/bm25/Cargo.toml:13), the seed being the 32
// bytes at offset 40 of the Meta tuple (tuples.rs:48-57).  Portable scalar code: a query has a handful of
// lexemes, this is nowhere near a hot path.  Checked against the published test vectors in tests/test_blake3.py.
#include "vbm25_internal.h"

#include <cstring>

namespace {

constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                            0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
constexpr uint8_t PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
constexpr uint32_t CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8, KEYED_HASH = 16;
constexpr size_t BLOCK = 64, CHUNK = 1024;

inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

inline void g(uint32_t *s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx;
    s[d] = rotr(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];
    s[b] = rotr(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my;
    s[d] = rotr(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];
    s[b] = rotr(s[b] ^ s[c], 7);
}

void compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len, uint32_t flags,
              uint32_t out[16]) {
    uint32_t s[16] = {cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7], IV[0], IV[1], IV[2], IV[3],
                      uint32_t(counter), uint32_t(counter >> 32), block_len, flags};
    uint32_t m[16];
    std::memcpy(m, block, sizeof m);
    for (int r = 0; r < 7; ++r) {
        g(s, 0, 4, 8, 12, m[0], m[1]);
        g(s, 1, 5, 9, 13, m[2], m[3]);
        g(s, 2, 6, 10, 14, m[4], m[5]);
        g(s, 3, 7, 11, 15, m[6], m[7]);
        g(s, 0, 5, 10, 15, m[8], m[9]);
        g(s, 1, 6, 11, 12, m[10], m[11]);
        g(s, 2, 7, 8, 13, m[12], m[13]);
        g(s, 3, 4, 9, 14, m[14], m[15]);
        uint32_t p[16];
        for (int i = 0; i < 16; ++i) p[i] = m[PERM[i]];
        std::memcpy(m, p, sizeof m);
    }
    for (int i = 0; i < 8; ++i) {
        s[i] ^= s[i + 8];
        s[i + 8] ^= cv[i];
    }
    std::memcpy(out, s, 64);
}

void words_from_le(const uint8_t *bytes, size_t n, uint32_t out[16]) {  // n <= 64, zero padded
    uint8_t buf[BLOCK] = {0};
    std::memcpy(buf, bytes, n);
    for (int i = 0; i < 16; ++i)
        out[i] = uint32_t(buf[4 * i]) | uint32_t(buf[4 * i + 1]) << 8 | uint32_t(buf[4 * i + 2]) << 16 | uint32_t(buf[4 * i + 3]) << 24;
}

// The node whose compression yields the next chaining value or, with ROOT, the output
struct Output {
    uint32_t cv[8], block[16];
    uint64_t counter;
    uint32_t block_len, flags;
    void chaining_value(uint32_t out[8]) const {
        uint32_t s[16];
        compress(cv, block, counter, block_len, flags, s);
        std::memcpy(out, s, 32);
    }
    void root_bytes(uint8_t *out, size_t n) const {
        uint64_t ctr = 0;
        while (n) {
            uint32_t s[16];
            compress(cv, block, ctr++, block_len, flags | ROOT, s);
            uint8_t b[64];
            for (int i = 0; i < 16; ++i) {
                b[4 * i] = uint8_t(s[i]);
                b[4 * i + 1] = uint8_t(s[i] >> 8);
                b[4 * i + 2] = uint8_t(s[i] >> 16);
                b[4 * i + 3] = uint8_t(s[i] >> 24);
            }
            const size_t take = n < 64 ? n : 64;
            std::memcpy(out, b, take);
            out += take;
            n -= take;
        }
    }
};

Output parent_output(const uint32_t left[8], const uint32_t right[8], const uint32_t key[8], uint32_t flags) {
    Output o;
    std::memcpy(o.cv, key, 32);
    std::memcpy(o.block, left, 32);
    std::memcpy(o.block + 8, right, 32);
    o.counter = 0;
    o.block_len = BLOCK;
    o.flags = PARENT | flags;
    return o;
}

}  // namespace

namespace vbm25 {

void blake3(const uint8_t *key32, const uint8_t *in, size_t len, uint8_t *out, size_t out_len) {
    uint32_t key[8];
    uint32_t flags = 0;
    if (key32) {
        for (int i = 0; i < 8; ++i)
            key[i] = uint32_t(key32[4 * i]) | uint32_t(key32[4 * i + 1]) << 8 | uint32_t(key32[4 * i + 2]) << 16 | uint32_t(key32[4 * i + 3]) << 24;
        flags = KEYED_HASH;
    } else {
        std::memcpy(key, IV, 32);
    }
    uint32_t stack[54][8];
    int stack_len = 0;
    uint64_t chunk_counter = 0;
    // every chunk but the last one is finished into a chaining value and merged up the tree
    Output last;
    size_t pos = 0;
    for (;;) {
        const size_t take = len - pos < CHUNK ? len - pos : CHUNK;
        const bool final_chunk = pos + take == len;
        // the chunk: blocks of 64 bytes chained through cv; the last block carries CHUNK_END
        uint32_t cv[8];
        std::memcpy(cv, key, 32);
        size_t off = 0;
        uint32_t blocks = 0;
        Output co;
        for (;;) {
            const size_t bl = take - off < BLOCK ? take - off : BLOCK;
            const bool last_block = off + bl == take;
            uint32_t words[16];
            words_from_le(in + pos + off, bl, words);
            const uint32_t f = flags | (blocks == 0 ? CHUNK_START : 0) | (last_block ? CHUNK_END : 0);
            if (last_block) {
                std::memcpy(co.cv, cv, 32);
                std::memcpy(co.block, words, 64);
                co.counter = chunk_counter;
                co.block_len = uint32_t(bl);
                co.flags = f;
                break;
            }
            uint32_t s[16];
            compress(cv, words, chunk_counter, BLOCK, f, s);
            std::memcpy(cv, s, 32);
            off += bl;
            ++blocks;
        }
        if (final_chunk) {
            last = co;
            break;
        }
        uint32_t ccv[8];
        co.chaining_value(ccv);
        ++chunk_counter;
        // merge completed subtrees: one pop per trailing zero bit of the number of chunks so far
        for (uint64_t total = chunk_counter; (total & 1) == 0; total >>= 1) {
            uint32_t pcv[8];
            parent_output(stack[stack_len - 1], ccv, key, flags).chaining_value(pcv);
            std::memcpy(ccv, pcv, 32);
            --stack_len;
        }
        std::memcpy(stack[stack_len++], ccv, 32);
        pos += take;
    }
    // fold the stack from the right: the root is the last parent (or the only chunk)
    Output o = last;
    while (stack_len > 0) {
        uint32_t rcv[8];
        o.chaining_value(rcv);
        o = parent_output(stack[--stack_len], rcv, key, flags);
    }
    o.root_bytes(out, out_len);
}

}  // namespace vbm25

extern "C" {

// intern (vector.rs:19-35)
int vbm25_intern(const uint8_t *seed32, const uint8_t *string, size_t len, uint8_t *key16) {
    if ((!string && len) || !key16) return vbm25::set_error(VBM25_ERR_INVALID, "NULL argument");
    const bool has_nul = len && std::memchr(string, 0, len) != nullptr;
    if (len < 16 && !has_nul) {
        std::memset(key16, 0, 16);
        if (len) std::memcpy(key16, string, len);
        return VBM25_OK;
    }
    if (!seed32) return vbm25::set_error(VBM25_ERR_INVALID, "a lexeme of 16 bytes or more needs the index's seed (Meta tuple)");
    uint8_t h[32];
    vbm25::blake3(seed32, string, len, h, 32);
    std::memcpy(key16, h, 16);
    if (key16[15] == 0) key16[15] = 1;
    return VBM25_OK;
}

int vbm25_blake3(const uint8_t *key32, const uint8_t *in, size_t len, uint8_t *out32) {
    if ((!in && len) || !out32) return vbm25::set_error(VBM25_ERR_INVALID, "NULL argument");
    vbm25::blake3(key32, in, len, out32, 32);
    return VBM25_OK;
}

}  // extern "C"
