// segment.cpp -- host side of libvbm25: sealed-segment construction.
//
// Produces the flattened index (include/vbm25.h, vbm25_index_desc) with the
// semantics of the reference's flush (crates/bm25/src/flush.rs:40-158):
//   * fieldnorm = length_to_fieldnorm(document length)      (bm25.rs:278-283)
//   * postings of a token cut into blocks of 128            (flush.rs:79-90)
//   * full blocks bit-packed (4-lane vertical layout, d1 deltas for doc ids),
//     tail blocks byte-packed                               (compression.rs:36-110)
//   * per block / per token WAND pair = first maximiser of tf()  (bm25.rs:297-332)
// Terms are encoded in parallel (one term per task) and stitched in key order;
// the output bytes do not depend on the number of threads.
//
// Also hosts the synthetic corpus generator of SURVEY section 8(d) and the
// algorithmic byte count used by bench.py's roofline line.

#include "vbm25_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <memory>
#include <numeric>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

namespace vbm25 {

// ---------------------------------------------------------------------------
// fieldnorm table: SmallFloat byte4ToInt with 24 exact values (== bm25.rs:15-272)
// ---------------------------------------------------------------------------
const uint32_t *fieldnorm_lengths() {
    static uint32_t table[256];
    static bool init = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            if (i < 24) {
                table[i] = i;
                continue;
            }
            uint32_t j = i - 24, mant = j & 7, ex = j >> 3;
            uint64_t v = ex == 0 ? mant : uint64_t(mant | 8) << (ex - 1);
            table[i] = uint32_t(v + 24);
        }
        return true;
    }();
    (void)init;
    return table;
}

uint8_t length_to_fieldnorm(uint32_t length) {
    const uint32_t *t = fieldnorm_lengths();
    int lo = 0, hi = 255;  // largest f with t[f] <= length
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (t[mid] <= length) lo = mid; else hi = mid - 1;
    }
    return uint8_t(lo);
}

void bm25_tables(uint32_t n_docs, uint64_t sum_len, double k1, double b, double *s1_256) {
    // Cache::new, bm25.rs:349-352 (the s1 table does not depend on the term)
    const double avgdl = double(sum_len) / double(n_docs);
    const uint32_t *t = fieldnorm_lengths();
    for (int f = 0; f < 256; ++f) s1_256[f] = k1 * (1.0 - b + b * double(t[f]) / avgdl);
}

double bm25_s0(uint32_t n_docs, uint32_t df, double k1) {
    // Cache::new, bm25.rs:348 with idf of bm25.rs:285-289; host libm log on purpose
    return std::log((double(n_docs) + 1.0) / (double(df) + 0.5)) * (k1 + 1.0);
}

namespace {

// ---------------------------------------------------------------------------
// Block encoder
// ---------------------------------------------------------------------------
inline uint32_t width_of(uint32_t ored) { return ored ? 32u - uint32_t(__builtin_clz(ored)) : 0u; }

// 128 fields of `b` bits -> 16*b bytes: four interleaved LSB-first lane streams.
void pack_lanes(const uint32_t *field, uint32_t b, uint8_t *dst) {
    uint32_t *out = reinterpret_cast<uint32_t *>(dst);  // dst is 8-byte aligned
    for (int lane = 0; lane < 4; ++lane) {
        uint64_t acc = 0;
        uint32_t fill = 0, w = 0;
        for (int step = 0; step < 32; ++step) {
            acc |= uint64_t(field[4 * step + lane]) << fill;
            fill += b;
            if (fill >= 32) {
                out[4 * w + lane] = uint32_t(acc);
                acc >>= 32;
                fill -= 32;
                ++w;
            }
        }
    }
}

struct TermOut {
    std::vector<uint8_t> blob;
    std::vector<uint32_t> min_doc, max_doc, wand_tf;
    std::vector<uint8_t> n, wand_fn, meta_doc, meta_tf;
    std::vector<uint32_t> len8;  // body length of each block in 8-byte units
    uint32_t df = 0;
    uint8_t term_wand_fn = 255;
    uint32_t term_wand_tf = 0;
    double term_wand_val = 0.0;
};

struct Encoder {
    const uint8_t *fieldnorm;
    double kp1;          // k1 + 1
    double denom[256];   // k1 * (1 - b + b * len(f) / avgdl), as in bm25.rs:291-295
    TermOut *out = nullptr;
    uint32_t docs[128], tfs[128];
    uint32_t fill = 0;

    void begin(TermOut *o) {
        out = o;
        fill = 0;
    }
    inline void push(uint32_t doc, uint32_t tf) {
        docs[fill] = doc;
        tfs[fill] = tf;
        if (++fill == 128) flush();
    }
    void finish() {
        if (fill) flush();
    }
    void flush() {
        const uint32_t n = fill;
        fill = 0;
        uint32_t delta[128];
        uint32_t or_d = 0, or_t = 0, prev = docs[0];
        for (uint32_t i = 0; i < n; ++i) {
            delta[i] = docs[i] - prev;
            prev = docs[i];
            or_d |= delta[i];
            or_t |= tfs[i];
        }
        const uint32_t bd = width_of(or_d), bt = width_of(or_t);
        uint8_t meta_d, meta_t;
        uint32_t len_d, len_t;
        if (n == 128) {
            meta_d = uint8_t(bd);
            meta_t = uint8_t(bt);
            len_d = 16 * bd;
            len_t = 16 * bt;
        } else {
            uint32_t wd = std::max(1u, (bd + 7) / 8), wt = std::max(1u, (bt + 7) / 8);
            meta_d = uint8_t(0x80 | wd);
            meta_t = uint8_t(0x80 | wt);
            len_d = wd * n;
            len_t = wt * n;
        }
        const uint32_t pad_d = (len_d + 7) & ~7u, pad_t = (len_t + 7) & ~7u;
        const size_t at = out->blob.size();
        out->blob.resize(at + pad_d + pad_t, 0);
        uint8_t *pd = out->blob.data() + at, *pt = pd + pad_d;
        if (n == 128) {
            if (bd == 32) std::memcpy(pd, docs, 512);  // raw absolute ids
            else if (bd) pack_lanes(delta, bd, pd);
            if (bt == 32) std::memcpy(pt, tfs, 512);
            else if (bt) pack_lanes(tfs, bt, pt);
        } else {
            const uint32_t wd = meta_d & 127, wt = meta_t & 127;
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t v = wd == 4 ? docs[i] : delta[i];
                for (uint32_t j = 0; j < wd; ++j) pd[i * wd + j] = uint8_t(v >> (8 * j));
                for (uint32_t j = 0; j < wt; ++j) pt[i * wt + j] = uint8_t(tfs[i] >> (8 * j));
            }
        }
        // block WAND pair: first maximiser, strict '<' (bm25.rs:311-318)
        double best = 0.0;
        uint8_t best_fn = 255;
        uint32_t best_tf = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint8_t f = fieldnorm[docs[i]];
            const double t = double(tfs[i]);
            const double v = (t * kp1) / (t + denom[f]);
            if (best < v) {
                best = v;
                best_fn = f;
                best_tf = tfs[i];
            }
        }
        if (out->term_wand_val < best) {  // Wand::extend, bm25.rs:319-325
            out->term_wand_val = best;
            out->term_wand_fn = best_fn;
            out->term_wand_tf = best_tf;
        }
        out->min_doc.push_back(docs[0]);
        out->max_doc.push_back(docs[n - 1]);
        out->n.push_back(uint8_t(n));
        out->wand_fn.push_back(best_fn);
        out->wand_tf.push_back(best_tf);
        out->meta_doc.push_back(meta_d);
        out->meta_tf.push_back(meta_t);
        out->len8.push_back((pad_d + pad_t) / 8);
        out->df += n;
    }
};

int resolve_threads(int threads) {
    if (threads > 0) return threads;
    unsigned hc = std::thread::hardware_concurrency();
    return hc ? int(hc) : 1;
}

// Runs fn(task, thread) over a pool.  No exception leaves a worker thread (that would be std::terminate,
// i.e. the death of the host process -- a PostgreSQL backend): the first one is kept and rethrown on the
// calling thread after every thread has been joined; thread creation failing is handled the same way.
template <class F>
void parallel_tasks(size_t n_tasks, int threads, F &&fn) {
    std::atomic<size_t> next{0};
    std::exception_ptr first_error;
    std::mutex error_mutex;
    auto worker = [&](int tid) {
        try {
            for (;;) {
                size_t i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= n_tasks) break;
                fn(i, tid);
            }
        } catch (...) {
            std::lock_guard<std::mutex> g(error_mutex);
            if (!first_error) first_error = std::current_exception();
            next.store(n_tasks, std::memory_order_relaxed);  // the others stop at their next task
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(size_t(threads > 1 ? threads - 1 : 0));  // (no reallocation -- no bad_alloc -- while joinable threads sit in it)
    try {
        for (int t = 1; t < threads; ++t) pool.emplace_back(worker, t);
    } catch (const std::system_error &) {
        // no more threads to be had (EAGAIN): not an error -- the ones that exist and this one drain the queue
    } catch (...) {  // anything else: stop the queue, join what runs, pass it on (a joinable std::thread destroyed = std::terminate)
        next.store(n_tasks, std::memory_order_relaxed);
        for (auto &th : pool) th.join();
        throw;
    }
    worker(0);
    for (auto &th : pool) th.join();
    if (first_error) std::rethrow_exception(first_error);
}

void init_encoder(Encoder &enc, const Segment &seg) {
    enc.fieldnorm = seg.doc_fieldnorm.data();
    enc.kp1 = seg.k1 + 1.0;
    bm25_tables(seg.n_docs, seg.sum_len, seg.k1, seg.b, enc.denom);
}

// Concatenate per-term outputs (already in key order) into the flat arrays.
void stitch(Segment &seg, std::vector<TermOut> &terms, const std::vector<uint32_t> &order,
            const uint8_t *keys16 /* indexed like `terms` */) {
    size_t n_terms = 0, n_blocks = 0, blob = 0;
    for (uint32_t t : order) {
        if (!terms[t].df) continue;
        ++n_terms;
        n_blocks += terms[t].n.size();
        blob += terms[t].blob.size();
    }
    seg.n_terms = uint32_t(n_terms);
    seg.n_blocks = uint32_t(n_blocks);
    seg.term_key.reserve(16 * n_terms);
    seg.term_df.reserve(n_terms);
    seg.term_wand_fn.reserve(n_terms);
    seg.term_wand_tf.reserve(n_terms);
    seg.term_first_block.reserve(n_terms + 1);
    seg.blk_min_doc.reserve(n_blocks);
    seg.blk_max_doc.reserve(n_blocks);
    seg.blk_n.reserve(n_blocks);
    seg.blk_wand_fn.reserve(n_blocks);
    seg.blk_wand_tf.reserve(n_blocks);
    seg.blk_meta_doc.reserve(n_blocks);
    seg.blk_meta_tf.reserve(n_blocks);
    seg.blk_off8.reserve(n_blocks + 1);
    seg.blob.reserve(blob);
    seg.term_first_block.push_back(0);
    seg.blk_off8.push_back(0);
    uint64_t off8 = 0;
    for (uint32_t t : order) {
        TermOut &o = terms[t];
        if (!o.df) continue;
        seg.term_key.insert(seg.term_key.end(), keys16 + 16ull * t, keys16 + 16ull * t + 16);
        seg.term_df.push_back(o.df);
        seg.term_wand_fn.push_back(o.term_wand_fn);
        seg.term_wand_tf.push_back(o.term_wand_tf);
        seg.blk_min_doc.insert(seg.blk_min_doc.end(), o.min_doc.begin(), o.min_doc.end());
        seg.blk_max_doc.insert(seg.blk_max_doc.end(), o.max_doc.begin(), o.max_doc.end());
        seg.blk_n.insert(seg.blk_n.end(), o.n.begin(), o.n.end());
        seg.blk_wand_fn.insert(seg.blk_wand_fn.end(), o.wand_fn.begin(), o.wand_fn.end());
        seg.blk_wand_tf.insert(seg.blk_wand_tf.end(), o.wand_tf.begin(), o.wand_tf.end());
        seg.blk_meta_doc.insert(seg.blk_meta_doc.end(), o.meta_doc.begin(), o.meta_doc.end());
        seg.blk_meta_tf.insert(seg.blk_meta_tf.end(), o.meta_tf.begin(), o.meta_tf.end());
        for (uint32_t l : o.len8) {
            off8 += l;
            seg.blk_off8.push_back(uint32_t(off8));
        }
        seg.blob.insert(seg.blob.end(), o.blob.begin(), o.blob.end());
        seg.term_first_block.push_back(uint32_t(seg.blk_n.size()));
        TermOut().blob.swap(o.blob);  // release early
    }
}

// ---------------------------------------------------------------------------
// Synthetic corpus
// ---------------------------------------------------------------------------
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    inline uint64_t next() {  // splitmix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    inline double unit() { return (double(next() >> 11) + 1.0) * (1.0 / 9007199254740992.0); }  // (0,1]
};
inline uint64_t mix(uint64_t a, uint64_t b, uint64_t c) {
    Rng r(a ^ (b * 0xD6E8FEB86659FD93ull) ^ (c * 0xCA5A826395121157ull));
    r.next();
    return r.next();
}

constexpr uint32_t DOC_CHUNK = 1u << 16;  // generation unit: (token, 65536-document chunk)

struct SynthCtx {
    uint32_t n_docs, vocab;
    uint64_t seed;
    std::vector<uint64_t> slot_start;  // n_docs + 1: prefix sum of draws per document
    std::vector<double> log1mp;        // per token: log(1 - p_t)
};

// Emit the postings of `token`: every draw slot independently is this token with
// probability p_t (geometric gap skipping); tf = hits inside one document.
template <class Emit>
void gen_token(const SynthCtx &cx, uint32_t token, Emit &&emit) {
    const double l1p = cx.log1mp[token];
    const uint64_t *S = cx.slot_start.data();
    for (uint32_t c0 = 0; c0 < cx.n_docs; c0 += DOC_CHUNK) {
        const uint32_t c1 = std::min<uint64_t>(cx.n_docs, uint64_t(c0) + DOC_CHUNK);
        Rng rng(mix(cx.seed, token, c0));
        const uint64_t end = S[c1];
        uint64_t pos = S[c0];
        uint32_t d = c0, cur_doc = UINT32_MAX, cur_tf = 0;
        for (;;) {
            double g = std::floor(std::log(rng.unit()) / l1p);
            if (!(g < 1e18)) break;
            pos += uint64_t(g);
            if (pos >= end) break;
            // slot -> document: interpolate inside the chunk, then walk
            if (S[d + 1] <= pos) {
                uint64_t span = end - S[c0];
                uint32_t guess = c0 + uint32_t((pos - S[c0]) * uint64_t(c1 - c0) / span);
                if (guess > d) d = guess;
                while (S[d] > pos) --d;
                while (S[d + 1] <= pos) ++d;
            }
            if (d == cur_doc) {
                ++cur_tf;
            } else {
                if (cur_tf) emit(cur_doc, cur_tf);
                cur_doc = d;
                cur_tf = 1;
            }
            ++pos;
        }
        if (cur_tf) emit(cur_doc, cur_tf);
    }
}

void write_key(uint32_t token, uint8_t *key16) {
    char buf[17];
    int n = std::snprintf(buf, sizeof buf, "%u", token);
    std::memset(key16, 0, 16);
    std::memcpy(key16, buf, size_t(n));
}

}  // namespace

// ---------------------------------------------------------------------------
// Segment methods
// ---------------------------------------------------------------------------
void Segment::desc(vbm25_index_desc *d) const {
    std::memset(d, 0, sizeof *d);
    d->n_docs = n_docs;
    d->n_terms = n_terms;
    d->n_blocks = n_blocks;
    d->sum_len = sum_len;
    d->blob_bytes = blob.size();
    d->k1 = k1;
    d->b = b;
    d->term_key = term_key.data();
    d->term_df = term_df.data();
    d->term_wand_fn = term_wand_fn.data();
    d->term_wand_tf = term_wand_tf.data();
    d->term_first_block = term_first_block.data();
    d->blk_min_doc = blk_min_doc.data();
    d->blk_max_doc = blk_max_doc.data();
    d->blk_n = blk_n.data();
    d->blk_wand_fn = blk_wand_fn.data();
    d->blk_wand_tf = blk_wand_tf.data();
    d->blk_meta_doc = blk_meta_doc.data();
    d->blk_meta_tf = blk_meta_tf.data();
    d->blk_off8 = blk_off8.data();
    d->blob = blob.data();
    d->doc_fieldnorm = doc_fieldnorm.data();
    d->doc_payload = doc_payload.data();
}

}  // namespace vbm25

using namespace vbm25;


// ---- save / load: little-endian dump of the arrays -------------------------
namespace {
constexpr uint64_t SEG_MAGIC = 0x31304745534D4276ull;  // "vBMSEG01"
template <class T>
bool put(FILE *f, const std::vector<T> &v) {
    uint64_t n = v.size();
    return fwrite(&n, 8, 1, f) == 1 && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <class T>
bool get(FILE *f, std::vector<T> &v, uint64_t file_bytes) {
    uint64_t n = 0;
    if (fread(&n, 8, 1, f) != 1) return false;
    if (n > file_bytes / sizeof(T)) return false;  // a length the file cannot hold: not a segment file
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}
}  // namespace


namespace vbm25 {
// Structural validation of a flattened segment (shared by vbm25_index_create and vbm25_segment_from_pages):
// what the reference would panic on as "data corruption" returns VBM25_ERR_CORRUPT.
int check_desc(const vbm25_index_desc *d) {
    if (!d) return set_error(VBM25_ERR_INVALID, "desc is NULL");
    if (!d->n_docs) {  // valid in the reference: every row is still in the growing segment; search returns nothing
        if (d->n_terms || d->n_blocks) return set_error(VBM25_ERR_CORRUPT, "terms or blocks without documents");
        return VBM25_OK;
    }
    if (!(d->k1 >= 1.2 && d->k1 <= 2.0) || !(d->b >= 0.0 && d->b <= 1.0))
        return set_error(VBM25_ERR_INVALID, "k1 must be in [1.2, 2] and b in [0, 1]");
    if (d->n_terms && (!d->term_key || !d->term_df || !d->term_first_block || !d->term_wand_tf || !d->term_wand_fn))
        return set_error(VBM25_ERR_INVALID, "term arrays missing");
    if (d->n_blocks && (!d->blk_min_doc || !d->blk_max_doc || !d->blk_n || !d->blk_meta_doc ||
                        !d->blk_meta_tf || !d->blk_off8 || !d->blob))
        return set_error(VBM25_ERR_INVALID, "block arrays missing");
    if (!d->doc_fieldnorm || !d->doc_payload)
        return set_error(VBM25_ERR_INVALID, "document arrays missing");
    if (d->n_terms) {
        if (d->term_first_block[0] != 0 || d->term_first_block[d->n_terms] != d->n_blocks)
            return set_error(VBM25_ERR_CORRUPT, "term_first_block does not cover the blocks");
        for (uint32_t t = 0; t < d->n_terms; ++t) {
            uint32_t nb = d->term_first_block[t + 1] - d->term_first_block[t];
            if (d->term_first_block[t + 1] < d->term_first_block[t] ||
                nb != (d->term_df[t] + 127) / 128 || d->term_df[t] == 0 || d->term_df[t] > d->n_docs)
                return set_error(VBM25_ERR_CORRUPT, "term %u: df / block count mismatch", t);
            if (t && std::memcmp(d->term_key + 16ull * (t - 1), d->term_key + 16ull * t, 16) >= 0)
                return set_error(VBM25_ERR_CORRUPT, "term keys not strictly ascending at %u", t);
        }
    } else if (d->n_blocks) {
        return set_error(VBM25_ERR_CORRUPT, "blocks without terms");
    }
    for (uint32_t t = 0; t < d->n_terms; ++t) {
        uint32_t b0 = d->term_first_block[t], b1 = d->term_first_block[t + 1];
        uint64_t cnt = 0;
        for (uint32_t j = b0; j < b1; ++j) {
            const uint32_t n = d->blk_n[j];
            const uint8_t md = d->blk_meta_doc[j], mt = d->blk_meta_tf[j];
            if (n < 1 || n > 128 || (j + 1 < b1 && n != 128))
                return set_error(VBM25_ERR_CORRUPT, "block %u: bad posting count %u", j, n);
            const bool full = n == 128;
            for (uint8_t mm : {md, mt}) {
                const uint32_t w = mm & 127;
                if (full ? ((mm >> 7) != 0 || w > 32) : ((mm >> 7) != 1 || w < 1 || w > 4))
                    return set_error(VBM25_ERR_CORRUPT, "block %u: bad codec metadata 0x%02x", j, mm);
            }
            const uint32_t ld = (md >> 7) ? (md & 127u) * n : 16u * (md & 127u);
            const uint32_t lt = (mt >> 7) ? (mt & 127u) * n : 16u * (mt & 127u);
            const uint64_t need = ((ld + 7) / 8) + ((lt + 7) / 8);
            if (d->blk_off8[j + 1] < d->blk_off8[j] || d->blk_off8[j + 1] - d->blk_off8[j] != need)
                return set_error(VBM25_ERR_CORRUPT, "block %u: body length mismatch", j);
            if (d->blk_min_doc[j] > d->blk_max_doc[j] || d->blk_max_doc[j] >= d->n_docs ||
                (j > b0 && d->blk_min_doc[j] <= d->blk_max_doc[j - 1]))
                return set_error(VBM25_ERR_CORRUPT, "block %u: document range out of order", j);
            cnt += n;
        }
        if (cnt != d->term_df[t]) return set_error(VBM25_ERR_CORRUPT, "term %u: df mismatch", t);
    }
    if (d->n_blocks && 8ull * d->blk_off8[d->n_blocks] > d->blob_bytes)
        return set_error(VBM25_ERR_CORRUPT, "blob shorter than the block offsets");
    return VBM25_OK;
}
}  // namespace vbm25

extern "C" {

int vbm25_segment_build(double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                        const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                        const uint64_t *term_start, const uint32_t *post_doc,
                        const uint32_t *post_tf, int threads, vbm25_segment **out) {
    if (!out) return set_error(VBM25_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!n_docs) return set_error(VBM25_ERR_INVALID, "segment without documents");
    if (!(k1 >= 1.2 && k1 <= 2.0) || !(b >= 0.0 && b <= 1.0))  // types.rs:18-45
        return set_error(VBM25_ERR_INVALID, "k1 must be in [1.2, 2] and b in [0, 1]");
    for (uint32_t t = 0; t + 1 < n_terms; ++t)
        if (std::memcmp(term_key + 16ull * t, term_key + 16ull * (t + 1), 16) >= 0)
            return set_error(VBM25_ERR_INVALID, "term keys must be strictly ascending");
    try {
        auto seg = std::make_unique<vbm25_segment>();
        seg->k1 = k1;
        seg->b = b;
        seg->n_docs = n_docs;
        seg->doc_fieldnorm.resize(n_docs);
        seg->doc_payload.assign(doc_payload, doc_payload + 3ull * n_docs);
        uint64_t sum = 0;
        for (uint32_t d = 0; d < n_docs; ++d) {
            sum += doc_len[d];
            seg->doc_fieldnorm[d] = length_to_fieldnorm(doc_len[d]);
        }
        seg->sum_len = sum;
        threads = resolve_threads(threads);
        std::vector<TermOut> outs(n_terms);
        std::atomic<int> bad{0};
        std::vector<Encoder> encs(static_cast<size_t>(threads));
        for (auto &e : encs) init_encoder(e, *seg);
        parallel_tasks(n_terms, threads, [&](size_t t, int tid) {
            Encoder &enc = encs[size_t(tid)];
            enc.begin(&outs[t]);
            uint32_t prev = 0;
            bool first = true;
            for (uint64_t p = term_start[t]; p < term_start[t + 1]; ++p) {
                uint32_t d = post_doc[p];
                if (d >= n_docs || post_tf[p] == 0 || (!first && d <= prev)) {
                    bad.store(1);
                    return;
                }
                prev = d;
                first = false;
                enc.push(d, post_tf[p]);
            }
            enc.finish();
        });
        if (bad.load())
            return set_error(VBM25_ERR_INVALID,
                             "mappings must be sorted by (token, document), ids < n_docs, tf > 0");
        std::vector<uint32_t> order(n_terms);
        std::iota(order.begin(), order.end(), 0u);
        stitch(*seg, outs, order, term_key);
        *out = seg.release();
        return VBM25_OK;
    } catch (const std::bad_alloc &) {
        return set_error(VBM25_ERR_NOMEM, "out of host memory while building segment");
    } catch (const std::exception &e) {  // nothing may unwind across the C ABI (std::system_error from std::thread, ...)
        return set_error(VBM25_ERR_INVALID, "internal error while building segment: %s", e.what());
    } catch (...) {
        return set_error(VBM25_ERR_INVALID, "internal error while building segment");
    }
}

int vbm25_segment_synth(const vbm25_synth_params *pr, vbm25_segment **out) {
    if (!pr || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (!pr->n_docs || !pr->vocab || !pr->mean_len)
        return set_error(VBM25_ERR_INVALID, "n_docs, vocab and mean_len must be positive");
    if (!(pr->k1 >= 1.2 && pr->k1 <= 2.0) || !(pr->b >= 0.0 && pr->b <= 1.0))
        return set_error(VBM25_ERR_INVALID, "k1 must be in [1.2, 2] and b in [0, 1]");
    try {
        const int threads = resolve_threads(pr->threads);
        auto seg = std::make_unique<vbm25_segment>();
        seg->k1 = pr->k1;
        seg->b = pr->b;
        seg->n_docs = pr->n_docs;
        SynthCtx cx;
        cx.n_docs = pr->n_docs;
        cx.vocab = pr->vocab;
        cx.seed = pr->seed;
        // draws per document
        cx.slot_start.resize(size_t(pr->n_docs) + 1);
        cx.slot_start[0] = 0;
        {
            Rng rng(mix(pr->seed, 0xD0C5, 0));
            const double mu = std::log(0.8 * double(pr->mean_len));
            for (uint32_t d = 0; d < pr->n_docs; ++d) {
                uint64_t len = pr->mean_len;
                if (pr->len_mode == 1) {
                    double u1 = rng.unit(), u2 = rng.unit();
                    double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
                    double v = std::nearbyint(std::exp(mu + 0.6 * z));
                    len = uint64_t(std::min(2000.0, std::max(8.0, v)));
                }
                cx.slot_start[d + 1] = cx.slot_start[d] + len;
            }
        }
        // token probabilities
        cx.log1mp.resize(pr->vocab);
        {
            double norm = 0.0;
            if (pr->zipf_s > 0)
                for (uint32_t t = 0; t < pr->vocab; ++t) norm += std::pow(double(t + 1), -pr->zipf_s);
            for (uint32_t t = 0; t < pr->vocab; ++t) {
                double p = pr->zipf_s > 0 ? std::pow(double(t + 1), -pr->zipf_s) / norm
                                          : 1.0 / double(pr->vocab);
                cx.log1mp[t] = std::log1p(-std::min(p, 0.999999));
            }
        }
        // tokens by descending expected size (load balance), keys for ordering
        std::vector<uint32_t> by_size(pr->vocab);
        std::iota(by_size.begin(), by_size.end(), 0u);
        if (pr->zipf_s <= 0) {
            // uniform: all the same size, keep natural order
        }  // zipf: token id == rank, already descending
        std::vector<uint8_t> keys(16ull * pr->vocab);
        for (uint32_t t = 0; t < pr->vocab; ++t) write_key(t, keys.data() + 16ull * t);

        // pass 1: document lengths = sum of tf over the generated postings (vector.rs:77-83)
        std::vector<std::atomic<uint32_t>> lens(pr->n_docs);
        for (auto &l : lens) l.store(0, std::memory_order_relaxed);
        parallel_tasks(pr->vocab, threads, [&](size_t i, int) {
            gen_token(cx, by_size[i], [&](uint32_t d, uint32_t tf) {
                lens[d].fetch_add(tf, std::memory_order_relaxed);
            });
        });
        seg->doc_fieldnorm.resize(pr->n_docs);
        seg->doc_payload.resize(3ull * pr->n_docs);
        uint64_t sum = 0;
        for (uint32_t d = 0; d < pr->n_docs; ++d) {
            uint32_t l = lens[d].load(std::memory_order_relaxed);
            sum += l;
            seg->doc_fieldnorm[d] = length_to_fieldnorm(l);
            uint32_t blk = d / 64;  // synthetic ctid, layout of fetcher.rs:218-225
            seg->doc_payload[3ull * d + 0] = uint16_t(blk >> 16);
            seg->doc_payload[3ull * d + 1] = uint16_t(blk & 0xffff);
            seg->doc_payload[3ull * d + 2] = uint16_t(d % 64 + 1);
        }
        seg->sum_len = sum;
        std::vector<std::atomic<uint32_t>>().swap(lens);

        // pass 2: regenerate and encode
        std::vector<TermOut> outs(pr->vocab);
        std::vector<Encoder> encs(static_cast<size_t>(threads));
        for (auto &e : encs) init_encoder(e, *seg);
        parallel_tasks(pr->vocab, threads, [&](size_t i, int tid) {
            uint32_t tok = by_size[i];
            Encoder &enc = encs[size_t(tid)];
            enc.begin(&outs[tok]);
            gen_token(cx, tok, [&](uint32_t d, uint32_t tf) { enc.push(d, tf); });
            enc.finish();
        });
        // key order = bytewise order of the decimal strings
        std::vector<uint32_t> order(pr->vocab);
        std::iota(order.begin(), order.end(), 0u);
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b2) {
            return std::memcmp(keys.data() + 16ull * a, keys.data() + 16ull * b2, 16) < 0;
        });
        seg->token_term.assign(pr->vocab, UINT32_MAX);
        {
            uint32_t id = 0;
            for (uint32_t t : order)
                if (outs[t].df) seg->token_term[t] = id++;
        }
        stitch(*seg, outs, order, keys.data());
        *out = seg.release();
        return VBM25_OK;
    } catch (const std::bad_alloc &) {
        return set_error(VBM25_ERR_NOMEM, "out of host memory while generating corpus");
    } catch (const std::exception &e) {
        return set_error(VBM25_ERR_INVALID, "internal error while generating corpus: %s", e.what());
    } catch (...) {
        return set_error(VBM25_ERR_INVALID, "internal error while generating corpus");
    }
}

int vbm25_segment_synth_token_terms(const vbm25_segment *seg, const uint32_t *tokens, uint32_t n,
                                    uint32_t *term_ids) {
    if (!seg || seg->token_term.empty())
        return set_error(VBM25_ERR_INVALID, "segment was not produced by vbm25_segment_synth");
    for (uint32_t i = 0; i < n; ++i)
        term_ids[i] = tokens[i] < seg->token_term.size() ? seg->token_term[tokens[i]] : UINT32_MAX;
    return VBM25_OK;
}

int vbm25_segment_desc(const vbm25_segment *seg, vbm25_index_desc *out) {
    if (!seg || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
    seg->desc(out);
    return VBM25_OK;
}

void vbm25_segment_free(vbm25_segment *seg) { delete seg; }

static int vbm25_segment_save_impl(const vbm25_segment *s, const char *path) {
    if (!s || !path) return set_error(VBM25_ERR_INVALID, "NULL argument");
    FILE *f = std::fopen(path, "wb");
    if (!f) return set_error(VBM25_ERR_INVALID, "cannot open %s for writing", path);
    uint64_t hdr[6] = {SEG_MAGIC, s->n_docs, s->n_terms, s->n_blocks, s->sum_len, 0};
    bool ok = fwrite(hdr, 8, 6, f) == 6 && fwrite(&s->k1, 8, 1, f) == 1 && fwrite(&s->b, 8, 1, f) == 1;
    ok = ok && put(f, s->term_key) && put(f, s->term_df) && put(f, s->term_wand_fn) &&
         put(f, s->term_wand_tf) && put(f, s->term_first_block) && put(f, s->blk_min_doc) &&
         put(f, s->blk_max_doc) && put(f, s->blk_n) && put(f, s->blk_wand_fn) &&
         put(f, s->blk_wand_tf) && put(f, s->blk_meta_doc) && put(f, s->blk_meta_tf) &&
         put(f, s->blk_off8) && put(f, s->blob) && put(f, s->doc_fieldnorm) &&
         put(f, s->doc_payload) && put(f, s->token_term);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? VBM25_OK : set_error(VBM25_ERR_INVALID, "short write to %s", path);
}

static int vbm25_segment_load_impl(const char *path, vbm25_segment **out) {
    if (!path || !out) return set_error(VBM25_ERR_INVALID, "NULL argument");
    *out = nullptr;
    FILE *f = std::fopen(path, "rb");
    if (!f) return set_error(VBM25_ERR_INVALID, "cannot open %s", path);
    auto s = std::make_unique<vbm25_segment>();
    std::fseek(f, 0, SEEK_END);
    const uint64_t fb = uint64_t(std::ftell(f));
    std::fseek(f, 0, SEEK_SET);
    uint64_t hdr[6];
    bool ok = fread(hdr, 8, 6, f) == 6 && hdr[0] == SEG_MAGIC && fread(&s->k1, 8, 1, f) == 1 &&
              fread(&s->b, 8, 1, f) == 1;
    if (ok) {
        s->n_docs = uint32_t(hdr[1]);
        s->n_terms = uint32_t(hdr[2]);
        s->n_blocks = uint32_t(hdr[3]);
        s->sum_len = hdr[4];
        ok = get(f, s->term_key, fb) && get(f, s->term_df, fb) && get(f, s->term_wand_fn, fb) &&
             get(f, s->term_wand_tf, fb) && get(f, s->term_first_block, fb) && get(f, s->blk_min_doc, fb) &&
             get(f, s->blk_max_doc, fb) && get(f, s->blk_n, fb) && get(f, s->blk_wand_fn, fb) &&
             get(f, s->blk_wand_tf, fb) && get(f, s->blk_meta_doc, fb) && get(f, s->blk_meta_tf, fb) &&
             get(f, s->blk_off8, fb) && get(f, s->blob, fb) && get(f, s->doc_fieldnorm, fb) &&
             get(f, s->doc_payload, fb) && get(f, s->token_term, fb);
    }
    std::fclose(f);
    if (!ok) return set_error(VBM25_ERR_CORRUPT, "%s is not a vbm25 segment file", path);
    *out = s.release();
    return VBM25_OK;
}

// The two tables every score goes through, as this library computes them (tests compare them with the reference's:
// tests/golden/fieldnorm_table.json is generated from bm25.rs:15-272).
int vbm25_fieldnorm_table(uint32_t *lengths256) {
    if (!lengths256) return set_error(VBM25_ERR_INVALID, "NULL argument");
    std::memcpy(lengths256, fieldnorm_lengths(), 256 * sizeof(uint32_t));
    return VBM25_OK;
}
int vbm25_cache_s1(uint32_t n_docs, uint64_t sum_len, double k1, double b, double *s1_256) {
    if (!s1_256) return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (n_docs == 0) return set_error(VBM25_ERR_INVALID, "no documents: the average length is 0 / 0");
    bm25_tables(n_docs, sum_len, k1, b, s1_256);
    return VBM25_OK;
}

uint64_t vbm25_query_bytes(const vbm25_index_desc *d, const uint32_t *term_ids, uint32_t n_terms,
                           uint32_t k) {
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n_terms; ++i) {
        uint32_t t = term_ids[i];
        if (t >= d->n_terms) continue;
        uint32_t b0 = d->term_first_block[t], b1 = d->term_first_block[t + 1];
        bytes += 8ull * (d->blk_off8[b1] - d->blk_off8[b0]) + 40ull * (b1 - b0) + d->term_df[t];
    }
    return bytes + 14ull * k;
}

// search.rs:83-135: the growing segment is scanned document by document before the WAND loop.
static int vbm25_growing_search_impl(const vbm25_index_desc *d, const uint8_t *query_keys, uint32_t n_keys, uint32_t k,
                         uint32_t n_grow, const uint64_t *g_start, const uint8_t *g_key, const uint32_t *g_tf,
                         const uint8_t *g_fieldnorm, const uint16_t *g_payload, const uint8_t *g_deleted,
                         vbm25_hit *hits, uint32_t *n_hits) {
    if (!d || !n_hits || (!hits && k) || (n_keys && !query_keys))
        return set_error(VBM25_ERR_INVALID, "NULL argument");
    if (k == 0) return set_error(VBM25_ERR_INVALID, "number of needed rows is set to 0");  // default.rs:114-116
    if (n_grow && (!g_start || !g_fieldnorm || !g_payload)) return set_error(VBM25_ERR_INVALID, "growing arrays missing");
    *n_hits = 0;
    for (uint32_t i = 1; i < n_keys; ++i)  // Query::checked_new, vector.rs:106-110
        if (std::memcmp(query_keys + 16ull * (i - 1), query_keys + 16ull * i, 16) >= 0)
            return set_error(VBM25_ERR_INVALID, "query keys must be strictly ascending");
    // tokens of the query that the sealed segment knows, with their Cache (search.rs:53-77)
    struct Tok {
        const uint8_t *key;
        double s0;
    };
    std::vector<Tok> toks;
    for (uint32_t i = 0; i < n_keys; ++i) {
        const uint8_t *key = query_keys + 16ull * i;
        uint32_t lo = 0, hi = d->n_terms;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (std::memcmp(d->term_key + 16ull * mid, key, 16) < 0) lo = mid + 1; else hi = mid;
        }
        if (lo < d->n_terms && !std::memcmp(d->term_key + 16ull * lo, key, 16))
            toks.push_back({key, bm25_s0(d->n_docs, d->term_df[lo], d->k1)});
    }
    double s1[256];
    bm25_tables(d->n_docs, d->sum_len, d->k1, d->b, s1);
    // Results (search.rs:284-314) restated as a sorted list: threshold = k-th score once k are held
    std::vector<vbm25_hit> top;
    top.reserve(k + 1);
    double threshold = 0.0;
    for (uint32_t g = 0; g < n_grow; ++g) {
        if (g_deleted && g_deleted[g]) continue;  // search.rs:113
        if (g_start[g + 1] < g_start[g]) return set_error(VBM25_ERR_INVALID, "g_start not monotone at %u", g);
        double result = 0.0;
        for (uint64_t p = g_start[g]; p < g_start[g + 1]; ++p) {
            const uint8_t *key = g_key + 16ull * p;
            size_t lo = 0, hi = toks.size();  // tokens.binary_search_by_key, search.rs:102,115
            while (lo < hi) {
                const size_t mid = (lo + hi) >> 1;
                if (std::memcmp(toks[mid].key, key, 16) < 0) lo = mid + 1; else hi = mid;
            }
            if (lo < toks.size() && !std::memcmp(toks[lo].key, key, 16)) {
                const double tf = double(g_tf[p]);
                result += (tf * toks[lo].s0) / (tf + s1[g_fieldnorm[g]]);  // Cache::evaluate, bm25.rs:355-358
            }
        }
        if (!(threshold < result)) continue;  // search.rs:121
        vbm25_hit h{};
        h.score = result;
        h.doc_id = UINT32_MAX - g;
        std::memcpy(h.payload, g_payload + 3ull * g, 6);
        // best first; among equal scores the earlier document stays ahead
        size_t pos = top.size();
        while (pos > 0 && top[pos - 1].score < result) --pos;
        top.insert(top.begin() + pos, h);
        if (top.size() > k) top.pop_back();
        if (top.size() == k) threshold = std::max(threshold, top.back().score);
    }
    std::copy(top.begin(), top.end(), hits);
    *n_hits = uint32_t(top.size());
    return VBM25_OK;
}

int vbm25_merge_hits(const vbm25_hit *sealed, uint32_t n_sealed, const vbm25_hit *grow, uint32_t n_grow,
                     uint32_t k, vbm25_hit *out, uint32_t *n_out) {
    if (!n_out || (!out && k) || (n_sealed && !sealed) || (n_grow && !grow))
        return set_error(VBM25_ERR_INVALID, "NULL argument");
    uint32_t i = 0, j = 0, n = 0;
    while (n < k && (i < n_sealed || j < n_grow)) {
        const bool take_grow = j < n_grow && (i >= n_sealed || grow[j].score > sealed[i].score);
        out[n++] = take_grow ? grow[j++] : sealed[i++];
    }
    *n_out = n;
    return VBM25_OK;
}


// evaluate.rs:22-74
int vbm25_evaluate(const vbm25_index_desc *d, const uint8_t *doc_key, const uint32_t *doc_tf, uint32_t n_doc,
                   const uint8_t *query_keys, uint32_t n_keys, double *score) {
    if (!d || !score || (n_doc && (!doc_key || !doc_tf)) || (n_keys && !query_keys))
        return set_error(VBM25_ERR_INVALID, "NULL argument");
    for (uint32_t i = 1; i < n_doc; ++i)  // Document::checked_new, vector.rs:56-61
        if (std::memcmp(doc_key + 16ull * (i - 1), doc_key + 16ull * i, 16) >= 0)
            return set_error(VBM25_ERR_INVALID, "document keys must be strictly ascending");
    for (uint32_t i = 1; i < n_keys; ++i)
        if (std::memcmp(query_keys + 16ull * (i - 1), query_keys + 16ull * i, 16) >= 0)
            return set_error(VBM25_ERR_INVALID, "query keys must be strictly ascending");
    uint64_t length = 0;  // Document::length, vector.rs:77-83: saturating
    for (uint32_t i = 0; i < n_doc; ++i) length = std::min<uint64_t>(length + doc_tf[i], UINT32_MAX);
    const uint8_t fieldnorm = length_to_fieldnorm(uint32_t(length));
    const double avgdl = double(d->sum_len) / double(d->n_docs);
    const double k1 = d->k1, b = d->b;
    const double document_length = double(fieldnorm_lengths()[fieldnorm]);
    size_t cursor = 0;
    double result = 0.0;
    for (uint32_t i = 0; i < n_keys; ++i) {
        const uint8_t *key = query_keys + 16ull * i;
        while (cursor < n_doc && std::memcmp(doc_key + 16ull * cursor, key, 16) < 0) ++cursor;
        if (!(cursor < n_doc && !std::memcmp(doc_key + 16ull * cursor, key, 16))) continue;
        uint32_t lo = 0, hi = d->n_terms;  // address_tokens::read
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (std::memcmp(d->term_key + 16ull * mid, key, 16) < 0) lo = mid + 1; else hi = mid;
        }
        if (!(lo < d->n_terms && !std::memcmp(d->term_key + 16ull * lo, key, 16))) continue;
        const double term_frequency = double(doc_tf[cursor]);
        const double idf = std::log((double(d->n_docs) + 1.0) / (double(d->term_df[lo]) + 0.5));  // bm25.rs:285-289
        const double tf = (term_frequency * (k1 + 1.0)) /
                          (term_frequency + k1 * (1.0 - b + b * document_length / avgdl));        // bm25.rs:291-295
        result += idf * tf;
    }
    *score = result;
    return VBM25_OK;
}

int vbm25_segment_load(const char *path, vbm25_segment **out) {
    return guarded([&] { return vbm25_segment_load_impl(path, out); });
}

int vbm25_growing_search(const vbm25_index_desc *d, const uint8_t *query_keys, uint32_t n_keys, uint32_t k,
                         uint32_t n_grow, const uint64_t *g_start, const uint8_t *g_key, const uint32_t *g_tf,
                         const uint8_t *g_fieldnorm, const uint16_t *g_payload, const uint8_t *g_deleted,
                         vbm25_hit *hits, uint32_t *n_hits) {
    return guarded([&] { return vbm25_growing_search_impl(d, query_keys, n_keys, k, n_grow, g_start, g_key, g_tf, g_fieldnorm, g_payload, g_deleted, hits, n_hits); });
}

int vbm25_segment_save(const vbm25_segment *s, const char *path) {
    return guarded([&] { return vbm25_segment_save_impl(s, path); });
}

}  // extern "C"
