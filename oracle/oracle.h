/*
 * oracle.h -- C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement of the reference's
 * Block-WAND query path (tensorchord/VectorChord-bm25, crates/bm25 + crates/simd
 * + crates/score).  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it; the product (vectorchord-bm25_amd/) never does.
 *
 * Parity status: the reference is Rust and cannot be compiled in this
 * environment (no rustc/cargo).  The restatement is pinned by
 *   - the reference's own golden id orderings (tests/sqllogictest, the .slt files),
 *   - its unit-test properties (codec round trips, Score bijection),
 *   - its differential-fuzz rule (Block-WAND == brute force, tests/fuzz:217-303).
 * What stays "parity unpinned": the order of equal-score hits (decided by Rust
 * std's BinaryHeap, which is not in the reference tree) and libm `log`.
 */
#ifndef VBM25_ORACLE_H
#define VBM25_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_index orc_index;

/* Result record, same layout as vbm25_hit in include/vbm25.h. */
typedef struct orc_hit {
    double score;
    uint32_t doc_id;
    uint16_t payload[3];
    uint16_t _pad;
} orc_hit;

/* Read-only view of the flattened index (format: DESIGN.md "Index layout"). */
typedef struct orc_index_view {
    uint32_t n_docs;
    uint32_t n_terms;
    uint32_t n_blocks;
    uint32_t _pad;
    uint64_t sum_len;
    uint64_t blob_bytes;
    double k1, b;
    const uint8_t *term_key;          /* n_terms x 16, ascending */
    const uint32_t *term_df;          /* n_terms */
    const uint8_t *term_wand_fn;      /* n_terms */
    const uint32_t *term_wand_tf;     /* n_terms */
    const uint32_t *term_first_block; /* n_terms + 1 */
    const uint32_t *blk_min_doc;      /* n_blocks */
    const uint32_t *blk_max_doc;
    const uint8_t *blk_n;             /* 1..128 */
    const uint8_t *blk_wand_fn;
    const uint32_t *blk_wand_tf;
    const uint8_t *blk_meta_doc;
    const uint8_t *blk_meta_tf;
    const uint32_t *blk_off8;         /* n_blocks + 1, units of 8 bytes into blob */
    const uint8_t *blob;
    const uint8_t *doc_fieldnorm;     /* n_docs */
    const uint16_t *doc_payload;      /* n_docs x 3 */
} orc_index_view;

/* ---- arithmetic (crates/bm25/src/bm25.rs) ---- */
uint32_t orc_fieldnorm_to_length(uint8_t fieldnorm);
uint8_t orc_length_to_fieldnorm(uint32_t length);
double orc_idf(uint32_t n_docs, uint32_t df);
double orc_tf(uint8_t fieldnorm, uint32_t tf, double k1, double b, double avgdl);
/* Cache::new + Cache::evaluate */
double orc_cache_evaluate(uint32_t n_docs, uint32_t df, double k1, double b, double avgdl,
                          uint8_t fieldnorm, uint32_t tf);

/* ---- score key (crates/score/src/lib.rs) ---- */
int64_t orc_score_from_f64(double v);
double orc_score_to_f64(int64_t s);

/* ---- codec (crates/bm25/src/compression.rs + crates/simd) ----
 * compress: returns metadata byte, writes payload to out (cap >= 512), *out_len.
 * n must be 1..128; n == 128 -> bit packing, else byte packing. */
uint8_t orc_compress_document_ids(uint32_t min_doc, const uint32_t *ids, uint32_t n,
                                  uint8_t *out, uint32_t *out_len);
uint8_t orc_compress_term_frequencies(const uint32_t *tfs, uint32_t n, uint8_t *out,
                                      uint32_t *out_len);
/* decompress: returns number of values written to out[128]. */
uint32_t orc_decompress_document_ids(uint32_t min_doc, uint8_t meta, const uint8_t *in,
                                     uint32_t in_len, uint32_t *out);
uint32_t orc_decompress_term_frequencies(uint8_t meta, const uint8_t *in, uint32_t in_len,
                                         uint32_t *out);

/* ---- Rust BinaryHeap model (assumed std semantics) ----
 * Runs a script on a max-heap of (key, tag) pairs ordered by key only.
 * ops[i] >= 0 : push (keys[i], tag i);  ops[i] == -1 : pop;  returns popped/sorted tags. */
uint32_t orc_heap_script(const int64_t *keys, const int32_t *ops, uint32_t n_ops,
                         int32_t *popped_tags, int32_t *sorted_tags, uint32_t *n_sorted);

/* ---- index construction (crates/bm25/src/flush.rs) ----
 * Segment = records (doc_len, payload) in doc-id order + mappings sorted by
 * (term key, doc id), given in CSR form over terms. */
orc_index *orc_index_build(double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                           const uint16_t *doc_payload /* n_docs x 3 */, uint32_t n_terms,
                           const uint8_t *term_key /* n_terms x 16 ascending */,
                           const uint64_t *term_start /* n_terms + 1 */,
                           const uint32_t *post_doc, const uint32_t *post_tf);
/* Adopt flattened arrays produced elsewhere (copied). */
orc_index *orc_index_from_view(const orc_index_view *view);
void orc_index_free(orc_index *);
void orc_index_get_view(const orc_index *, orc_index_view *out);

/* ---- query path ----
 * A query is a strictly ascending list of term ranks (position of the key in
 * term_key order); ranks >= n_terms mean "token not in index" and are ignored,
 * as search.rs:59-61 ignores unknown keys.  Each returns the number of hits. */

/* Faithful restatement of bm25::search (search.rs:28-282), sealed segment,
 * filter == true, empty growing segment. */
uint32_t orc_search_wand(const orc_index *, const uint32_t *terms, uint32_t n_terms, uint32_t k,
                         orc_hit *out);
/* Canonical brute force: every document scored with Cache::evaluate summed in
 * ascending key order; ordered by (score desc, doc id asc). */
/* blocks the calling thread's orc_search_wand calls have decompressed (fill_block, search.rs:498-518) since the last reset */
unsigned long long orc_wand_blocks_decoded(int reset);
uint32_t orc_search_brute(const orc_index *, const uint32_t *terms, uint32_t n_terms, uint32_t k,
                          orc_hit *out);
/* Batch drivers (one query per thread, PostgreSQL's execution model).
 * out: nq x k hits, n_hits: nq.  mode 0 = wand, 1 = brute.  Returns seconds. */
double orc_search_batch(const orc_index *, const uint32_t *terms, const uint32_t *q_off,
                        uint32_t nq, uint32_t k, int mode, int threads, orc_hit *out,
                        uint32_t *n_hits);
/* Growing-segment scan of search.rs:83-135 merged in front of WAND: docs given
 * as CSR (term rank, tf) lists with fieldnorm + payload + deleted flag. */
uint32_t orc_search_wand_growing(const orc_index *, const uint32_t *terms, uint32_t n_terms,
                                 uint32_t k, uint32_t n_grow, const uint64_t *g_start,
                                 const uint32_t *g_term, const uint32_t *g_tf,
                                 const uint8_t *g_fieldnorm, const uint16_t *g_payload,
                                 const uint8_t *g_deleted, orc_hit *out);
/* bm25::evaluate (evaluate.rs:22-74): one document against one query;
 * doc given as ascending (term rank, tf) pairs. Returns the Score key. */
int64_t orc_evaluate(const orc_index *, const uint32_t *doc_terms, const uint32_t *doc_tfs,
                     uint32_t n_doc_terms, const uint32_t *terms, uint32_t n_terms);

/* ---- the reference's on-disk layout (oracle/pages.cpp; parity unpinned, see its header) ----
 * A relation of PostgreSQL 8 KiB pages written the way build.rs:22-71 / flush.rs:40-158 /
 * insert.rs:23-79 write it: Meta, documents / tokens / summaries / blocks tapes, address trees,
 * vectors tape (growing segment), Jump.  Used to test the product's page reader. */
typedef struct orc_pages orc_pages;
orc_pages *orc_pages_build(const orc_index *, const uint8_t *seed32 /* may be NULL */);
void orc_pages_insert(orc_pages *, const uint16_t *payload3, uint32_t n_elem, const uint8_t *keys,
                      const uint32_t *tfs);
void orc_pages_mark_deleted_growing(orc_pages *, uint32_t nth);
uint32_t orc_pages_count(const orc_pages *);
const uint8_t *orc_pages_get(const orc_pages *, uint32_t page_id);
uint8_t *orc_pages_get_mut(orc_pages *, uint32_t page_id);
void orc_pages_free(orc_pages *);

/* Scalar model of the device's dense-window kernel (dense_model.inc): windows [lo, hi) of at most wmax documents
 * (first window w0, doubling), order-free 16-bit fixed-point upper-bound sums, MaxScore split + block-max test of the
 * non-essential terms' blocks (phases: 0 off, 1 the kernel's rule -- head terms in one phase --, 2 a phase per term,
 * 3 runs of one document-frequency class), histogram threshold, candidates re-scored exactly at a flush.  A test aid
 * for the kernel's bounds, not an oracle.  stats8: windows, blocks, blocks fetched untested, blocks tested / skipped,
 * candidates buffered / re-scored exactly, phases.  Returns 0xffffffff if an accumulator left its 16 bits. */
uint32_t orc_dense_model(const orc_index *, const uint32_t *terms, uint32_t n_terms, uint32_t k, uint32_t wmax,
                         uint32_t w0, uint32_t lo, uint32_t hi, int phases, orc_hit *out, uint64_t *stats8);
/* Algorithmic bytes of one query per SURVEY section 8(d). */
uint64_t orc_query_bytes(const orc_index *, const uint32_t *terms, uint32_t n_terms, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif
