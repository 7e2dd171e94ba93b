/*
 * vbm25.h -- C ABI of the MI355X-native BM25 top-k scorer (libvbm25.so).
 *
 * Drop-in boundary: this library replaces ONE call of the reference,
 *     bm25::search(&index, k, &query, filter)
 * made at /root/reference/src/index/bm25/scanners/default.rs:117-129 and defined
 * at crates/bm25/src/search.rs:28-36, plus the index flattening that feeds it.
 * INTEGRATION.md shows the Rust `extern "C"` block and the replacement body of
 * DefaultBuilder::build a maintainer would add.
 *
 * Conventions: plain pointers and sizes, caller-owned
 * outputs, no callbacks, no exceptions or longjmp across the boundary.  Every
 * function returns 0 on success or a negative vbm25_status; the message for the
 * last failure on the calling thread is available from vbm25_last_error().
 * One handle may be used from one thread at a time.
 */
#ifndef VBM25_H
#define VBM25_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vbm25_status {
    VBM25_OK = 0,
    VBM25_ERR_INVALID = -1,     /* bad argument (reference: pgrx::error! at default.rs:114-116 etc.) */
    VBM25_ERR_CORRUPT = -2,     /* index arrays inconsistent (reference: panic "data corruption") */
    VBM25_ERR_DEVICE = -3,      /* HIP runtime failure / no gfx950 device */
    VBM25_ERR_UNSUPPORTED = -4, /* valid request outside what the GPU path covers */
    VBM25_ERR_NOMEM = -5
} vbm25_status;

/* One result: replaces (Reverse<Score>, AlwaysEqual<[u16; 3]>) of search.rs:33.
 * score is the positive f64 (the SQL operator negates it, operators.rs:54);
 * payload is the heap ctid key (fetcher.rs:218-232). */
typedef struct vbm25_hit {
    double score;
    uint32_t doc_id;
    uint16_t payload[3];
    uint16_t _pad;
} vbm25_hit;

/* Flattened sealed segment: the values flush.rs:40-158 writes to the Token /
 * Summary / Block / Document tapes (tuples.rs:756-1069), as arrays.
 * All pointers are host memory, borrowed for the duration of the call. */
typedef struct vbm25_index_desc {
    uint32_t n_docs;              /* JumpTuple.number_of_documents */
    uint32_t n_terms;
    uint32_t n_blocks;
    uint32_t _pad;
    uint64_t sum_len;             /* JumpTuple.sum_of_document_lengths */
    uint64_t blob_bytes;
    double k1, b;                 /* MetaTuple.k1 / b */
    const uint8_t *term_key;          /* n_terms x 16, ascending (TokenTuple.id) */
    const uint32_t *term_df;          /* TokenTuple.number_of_documents */
    const uint8_t *term_wand_fn;      /* TokenTuple.wand_fieldnorm */
    const uint32_t *term_wand_tf;     /* TokenTuple.wand_term_frequency */
    const uint32_t *term_first_block; /* n_terms + 1; summaries of a token are contiguous */
    const uint32_t *blk_min_doc;      /* SummaryTuple.min_document_id */
    const uint32_t *blk_max_doc;      /* SummaryTuple.max_document_id */
    const uint8_t *blk_n;             /* SummaryTuple.number_of_documents (1..128) */
    const uint8_t *blk_wand_fn;       /* SummaryTuple.wand_fieldnorm */
    const uint32_t *blk_wand_tf;      /* SummaryTuple.wand_term_frequency */
    const uint8_t *blk_meta_doc;      /* BlockTuple.metadata_document_ids */
    const uint8_t *blk_meta_tf;       /* BlockTuple.metadata_term_frequencies */
    const uint32_t *blk_off8;         /* n_blocks + 1: block body offset in blob, units of 8 B */
    const uint8_t *blob;              /* per block: doc-id bytes, pad to 8, tf bytes, pad to 8 */
    const uint8_t *doc_fieldnorm;     /* DocumentTuple.fieldnorm, n_docs */
    const uint16_t *doc_payload;      /* DocumentTuple.payload, n_docs x 3 */
} vbm25_index_desc;

const char *vbm25_last_error(void);
const char *vbm25_version(void);

/* ------------------------------------------------------------------------
 * Host side: sealed-segment construction (replaces flush.rs:40-158 for the
 * benchmark / test harness; CPU, multi-threaded, byte-identical output).
 * ---------------------------------------------------------------------- */
typedef struct vbm25_segment vbm25_segment;

/* Segment = records (length, payload) in doc-id order + mappings sorted by
 * (token key, doc id) in CSR form (segment.rs:19-50). threads <= 0: all cores. */
int vbm25_segment_build(double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                        const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                        const uint64_t *term_start, const uint32_t *post_doc,
                        const uint32_t *post_tf, int threads, vbm25_segment **out);

/* The same construction on the device (csrc/flush.hip): one wave per 128-posting block (bit
 * width = OR of the deltas, 4-lane vertical packing, first-maximiser WAND pairs).  Same arguments, same
 * vbm25_segment, byte for byte; needs a gfx950 device (VBM25_ERR_DEVICE otherwise -- no silent fallback). */
int vbm25_segment_build_device(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                               const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                               const uint64_t *term_start, const uint32_t *post_doc,
                               const uint32_t *post_tf, vbm25_segment **out);

/* The same from mappings in ANY order -- (token rank, document, term frequency) triples as the tokenizer emits them:
 * the device radix-sorts them into the (token, document) order of segment.rs:41-45 (the reference merges sorted
 * runs, io.rs:244-282) and encodes.  Byte-identical to vbm25_segment_build on the sorted CSR form.  Errors: a token
 * rank >= n_terms, a token without mappings, a repeated (token, document) pair, tf = 0 (VBM25_ERR_INVALID). */
int vbm25_segment_build_device_unsorted(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                                        const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                                        uint64_t n_mappings, const uint32_t *map_term, const uint32_t *map_doc,
                                        const uint32_t *map_tf, vbm25_segment **out);

/* Synthetic corpus of SURVEY section 8(d), generated per token so that 10M-50M
 * documents stream: every document is `len` i.i.d. token draws (uniform, or
 * Zipf(s) over token rank when zipf_s > 0); token t's key is its ASCII decimal,
 * zero padded (vector.rs:21-24 short path).  len_mode 0: every document has
 * mean_len draws; 1: clamp(round(LogNormal(ln(0.8*mean_len), 0.6)), 8, 2000). */
typedef struct vbm25_synth_params {
    uint32_t n_docs;
    uint32_t vocab;
    uint32_t mean_len;
    uint32_t len_mode;
    double zipf_s;
    double k1, b;
    uint64_t seed;
    int threads;
    int _pad;
} vbm25_synth_params;
int vbm25_segment_synth(const vbm25_synth_params *params, vbm25_segment **out);
/* token number -> term id (rank of its key among the keys present), UINT32_MAX if absent */
int vbm25_segment_synth_token_terms(const vbm25_segment *, const uint32_t *tokens, uint32_t n,
                                    uint32_t *term_ids);

int vbm25_segment_desc(const vbm25_segment *, vbm25_index_desc *out);
void vbm25_segment_free(vbm25_segment *);
/* Serialise / load the flattened arrays (bench.py: rank 0 builds, other ranks load). */
int vbm25_segment_save(const vbm25_segment *, const char *path);
int vbm25_segment_load(const char *path, vbm25_segment **out);

/* Algorithmic bytes of one query, SURVEY section 8(d) (exhaustive evaluation:
 * block bodies + 16 B block header + 24 B summary per block + 1 fieldnorm byte
 * per posting + 14 B per returned hit). term ids >= n_terms are ignored. */
uint64_t vbm25_query_bytes(const vbm25_index_desc *, const uint32_t *term_ids, uint32_t n_terms,
                           uint32_t k);

/* FIELDNORM_TO_LENGTH (bm25.rs:15-272; 256 entries) and the term-independent half of Cache::new, s1[f] = k1 * (1 - b +
 * b * FIELDNORM_TO_LENGTH[f] / avgdl) with avgdl = sum_len / n_docs (bm25.rs:349-352), exactly as the library uses them. */
int vbm25_fieldnorm_table(uint32_t *lengths256);
int vbm25_cache_s1(uint32_t n_docs, uint64_t sum_len, double k1, double b, double *s1_256);

/* ------------------------------------------------------------------------
 * Host side of the shim: the growing (unsealed) segment, search.rs:83-135.
 * Documents inserted since the last VACUUM live in VectorTuples, not in
 * posting lists; the reference scores every one of them on the CPU before the
 * WAND loop, and so does the shim -- these two functions are that code.
 *
 * vbm25_growing_search: `query_keys` = the Query's sorted 16-byte keys; keys
 * without a TokenTuple are dropped (search.rs:59-61) and the others get the
 * sealed segment's statistics (Cache::new, search.rs:69-75).  Documents are
 * given as CSR over their elements (VectorTuple `Element{key, value}`, in
 * tuple order): score = sum of Cache::evaluate(fieldnorm, tf) over the elements
 * whose key is in the query, in element order; deleted documents are skipped; a
 * document enters only if threshold < score (so zero-score documents never
 * do, and a document tying the k-th score does not replace it).  Output: at
 * most k hits, best first, ties by document order; doc_id = 0xFFFFFFFF - index
 * of the document in the growing list (not a sealed document id).
 *
 * vbm25_merge_hits: top-k of the union of two best-first lists (sealed hits from
 * the device, growing hits from above); on equal scores sealed hits come first
 * (the reference leaves ties to BinaryHeap; unpinned).
 * ---------------------------------------------------------------------- */
int vbm25_growing_search(const vbm25_index_desc *desc, const uint8_t *query_keys, uint32_t n_keys,
                         uint32_t k, uint32_t n_grow, const uint64_t *g_start, const uint8_t *g_key,
                         const uint32_t *g_tf, const uint8_t *g_fieldnorm, const uint16_t *g_payload,
                         const uint8_t *g_deleted, vbm25_hit *hits, uint32_t *n_hits);
int vbm25_merge_hits(const vbm25_hit *sealed, uint32_t n_sealed, const vbm25_hit *grow, uint32_t n_grow,
                     uint32_t k, vbm25_hit *out, uint32_t *n_out);

/* bm25::evaluate (evaluate.rs:22-74): one document against one query with the sealed segment's
 * statistics -- what `tsvector <&> bm25query` computes when it runs as a plain function
 * (operators.rs:48-54, which returns the negated value).  Host code, as in the reference.  The
 * document's elements (key, tf) must be in ascending key order (vector.rs:50-75); its length is the
 * saturating sum of the tfs; result = sum over the query's keys found in both the document and the
 * index of idf(N, df) * tf(fieldnorm, tf, k1, b, avgdl), in key order (bm25.rs:285-295). */
int vbm25_evaluate(const vbm25_index_desc *desc, const uint8_t *doc_key, const uint32_t *doc_tf,
                   uint32_t n_doc_elements, const uint8_t *query_keys, uint32_t n_keys, double *score);

/* ------------------------------------------------------------------------
 * Host side: reading a bm25 index relation in the reference's own on-disk
 * format (PostgreSQL 8 KiB pages), the step between PostgreSQL and the GPU.
 *
 * `read_page(ctx, page_id)` returns the 8192-byte image of one page of the index
 * relation (the shim wraps ReadBuffer / a snapshot of the relation file), or
 * NULL.  Page layout: src/index/storage.rs:49-170 over PostgreSQL's page header
 * (24 B) + 4-byte line pointers (slots are 1-based) + 8-byte special area
 * Opaque{next, flags} (crates/bm25/src/lib.rs:41-46).
 *
 * vbm25_segment_from_pages walks what search() reads, in the order maintain.rs
 * :104-161 walks it: Meta (page 0, slot 1: magic "vchordbm", version 1, k1, b,
 * ptr_jump) -> Jump -> documents tape -> tokens tape -> summaries tape ->
 * blocks tape (tuples.rs:48-94,141-203,756-781,833-862,900-934,973-1025), and
 * checks while flattening that every token's summaries and every summary's
 * block sit where the WAND pointers say.  The result is an ordinary
 * vbm25_segment: vbm25_segment_desc + vbm25_index_create put it on the GPU
 * with no re-encoding (block bodies are copied byte for byte).  Anything the
 * reference would panic on ("data corruption", bad magic / version) returns
 * VBM25_ERR_CORRUPT.
 *
 * vbm25_growing_from_pages collects the unsealed documents of the same relation
 * (vectors tape from Jump.ptr_vectors; VectorTuple _2 / _1 / _0, tuples.rs
 * :326-426, state machine of search.rs:83-135) in the CSR form
 * vbm25_growing_search takes.
 * ---------------------------------------------------------------------- */
typedef const uint8_t *(*vbm25_read_page_fn)(void *ctx, uint32_t page_id);
int vbm25_segment_from_pages(vbm25_read_page_fn read_page, void *ctx, vbm25_segment **out);

typedef struct vbm25_growing vbm25_growing;
typedef struct vbm25_growing_desc {
    uint32_t n_docs;
    uint32_t _pad;
    uint64_t n_elements;
    const uint64_t *start;     /* n_docs + 1 */
    const uint8_t *key;        /* n_elements x 16 */
    const uint32_t *tf;        /* n_elements */
    const uint8_t *fieldnorm;  /* n_docs */
    const uint16_t *payload;   /* n_docs x 3 */
    const uint8_t *deleted;    /* n_docs */
} vbm25_growing_desc;
int vbm25_growing_from_pages(vbm25_read_page_fn read_page, void *ctx, vbm25_growing **out);
/* Cache key of the HBM copy of a relation's sealed segment: 32 bytes hashed (BLAKE3) over the Meta and Jump
 * tuples.  VACUUM replaces the sealed segment by rewriting the Jump tuple (maintain.rs:268-298: new tape
 * pointers, document count, sum of lengths), REINDEX rewrites Meta (new seed): either changes the fingerprint,
 * and the shim rebuilds its vbm25_index.  Inserts only append to the vectors tape and leave it unchanged
 * (the growing segment is read per query). */
int vbm25_pages_fingerprint(vbm25_read_page_fn read_page, void *ctx, uint8_t *out32);
/* MetaTuple.seed (tuples.rs:48-57): the key of vbm25_intern's hash for this index. */
int vbm25_pages_seed(vbm25_read_page_fn read_page, void *ctx, uint8_t *seed32);

/* intern (vector.rs:19-35): a lexeme -> its 16-byte token key.  Shorter than 16 bytes and without NUL: the
 * bytes, zero padded (seed32 may be NULL).  Otherwise the first 16 bytes of blake3::keyed_hash(seed, lexeme)
 * (blake3 1.8.4; implemented in csrc/blake3.cpp from the specification), last byte forced non-zero. */
int vbm25_intern(const uint8_t *seed32, const uint8_t *string, size_t len, uint8_t *key16);
int vbm25_growing_get_desc(const vbm25_growing *, vbm25_growing_desc *out);
void vbm25_growing_free(vbm25_growing *);

/* ------------------------------------------------------------------------
 * Device side
 * ---------------------------------------------------------------------- */
typedef struct vbm25_index vbm25_index; /* owns the HBM copy of one sealed segment */
typedef struct vbm25_batch vbm25_batch; /* owns query / result buffers for one batch shape */

/* Validates the arrays, uploads them to HBM on `device` (HIP ordinal) and
 * derives the GPU-side structures (per-posting fieldnorm stream, per-term s0,
 * the shared s1[256] table of bm25.rs:340-354). */
int vbm25_index_create(const vbm25_index_desc *desc, int device, vbm25_index **out);
void vbm25_index_destroy(vbm25_index *);
/* HBM bytes held by the index. */
uint64_t vbm25_index_device_bytes(const vbm25_index *);

/* address_tokens::read (address_tokens.rs:61-98) for n keys at once: term id of
 * each 16-byte key, or UINT32_MAX when the token is not in the index (such
 * tokens are ignored by search, search.rs:59-61). */
int vbm25_lookup_terms(const vbm25_index *, const uint8_t *keys, uint32_t n, uint32_t *term_ids);

/* bm25::search for nq queries at once (filter == true, sealed segment only).
 * Query q is term_ids[q_off[q] .. q_off[q+1]), strictly ascending (Query::new,
 * vector.rs:101-110); ids >= n_terms are ignored.  k = bm25.limit (1..=65535,
 * gucs.rs:37-46); k == 0 -> VBM25_ERR_INVALID like default.rs:114-116.
 * hits: nq x k, caller owned; n_hits: nq.  Results per query are best first:
 * score descending, ties by ascending doc id.  Synchronous. */
int vbm25_search_batch(vbm25_index *, const uint32_t *term_ids, const uint32_t *q_off,
                       uint32_t nq, uint32_t k, vbm25_hit *hits, uint32_t *n_hits);

/* Same computation with the batch resident in HBM: create once, upload
 * queries, run (asynchronous on `hip_stream`, NULL = default stream), fetch. */
int vbm25_batch_create(vbm25_index *, uint32_t max_queries, uint32_t max_total_terms, uint32_t k,
                       vbm25_batch **out);
void vbm25_batch_destroy(vbm25_batch *);
int vbm25_batch_set_queries(vbm25_batch *, const uint32_t *term_ids, const uint32_t *q_off,
                            uint32_t nq);
int vbm25_batch_run(vbm25_batch *, void *hip_stream);
int vbm25_batch_fetch(vbm25_batch *, vbm25_hit *hits, uint32_t *n_hits);
/* Device address of the nq x k vbm25_hit array / the nq counts (valid after run; a batch object whose device
 * addresses were asked for leaves complete records there after EVERY run -- without this call a run may leave a
 * query to vbm25_batch_fetch, which then repeats the scan for it before it returns). */
int vbm25_batch_device_results(vbm25_batch *, void **hits, void **n_hits);
/* When enabled, run() brackets the posting-scan kernel with HIP events on the
 * launch stream; kernel_ms() synchronises and returns the average duration of
 * the launches recorded since the last call. */
int vbm25_batch_set_timing(vbm25_batch *, int enabled);
int vbm25_batch_kernel_ms(vbm25_batch *, double *avg_ms, uint32_t *n_launches);

/* The same boundary PIPELINED (the caller hands over host buffers and gets host buffers back, as bm25::search
 * returns a Vec, search.rs:28-36): up to `depth` batches are in flight at once, each on its own stream with its
 * own pinned staging -- the upload of batch n + 1 and the records of batch n - 1 (written straight into pinned memory
 * by the last kernel of its scan) overlap the scan of batch n, and the host never waits for the device between a
 * submit and the matching collect.  One host thread drives a stream object (two threads: two objects).
 *   vbm25_stream_submit   copies the queries into pinned memory and enqueues upload and scan; returns at
 *                         once.  VBM25_ERR_INVALID when `depth` batches are already in flight (collect first).
 *   vbm25_stream_collect  waits for the OLDEST batch in flight and writes its records (nq x k hits, nq counts, in
 *                         submission order -- first in, first out); *nq_out = its number of queries.
 *                         VBM25_ERR_INVALID when nothing is in flight.
 * Records are byte-identical to vbm25_search_batch's. */
typedef struct vbm25_stream vbm25_stream;
int vbm25_stream_create(vbm25_index *, uint32_t depth, uint32_t max_queries, uint32_t max_total_terms, uint32_t k,
                        vbm25_stream **out);
void vbm25_stream_destroy(vbm25_stream *);
int vbm25_stream_submit(vbm25_stream *, const uint32_t *term_ids, const uint32_t *q_off, uint32_t nq);
int vbm25_stream_collect(vbm25_stream *, vbm25_hit *hits, uint32_t *n_hits, uint32_t *nq_out);
int vbm25_stream_in_flight(const vbm25_stream *);

/* bm25::evaluate (evaluate.rs:22-74) for n_docs documents against ONE query on the device: the seq-scan
 * form of `tsvector <&> bm25query` (src/index/operators.rs:22-55), batched.  Everything is in term-id space
 * (vbm25_lookup_terms): q_terms = the query's ids, strictly ascending; ids >= the index's term count (tokens
 * that are not in the index) are ignored.  Document i is the elements doc_start[i] .. doc_start[i+1]: term id
 * (ascending in key order; UINT32_MAX for a key that is not in the index -- it still counts for the document's
 * length) and term frequency (> 0).  scores[i] = sum of idf * tf in query key order, bit-identical to
 * vbm25_evaluate / the reference (idf through the host's libm log).  The SQL operator negates the value. */
int vbm25_evaluate_batch(vbm25_index *, const uint32_t *q_terms, uint32_t n_q_terms, uint32_t n_docs,
                         const uint64_t *doc_start, const uint32_t *doc_term, const uint32_t *doc_tf,
                         double *scores);

/* ------------------------------------------------------------------------
 * Device-resident build (SURVEY 8(f)-1 without the round trip): the sealed segment is encoded in HBM and STAYS there
 * (vbm25_device_segment); vbm25_index_create_from_device makes the index of it with device-to-device copies and
 * device-side derivation -- no posting crosses the PCIe link.  (flush.rs:40-158, io.rs:244-282.)
 *
 * vbm25_device_segment_build: arguments and result of vbm25_segment_build_device, kept on `device`.
 * vbm25_device_segment_synth: the synthetic corpus of vbm25_segment_synth GENERATED on the device (same model, same
 * counter-based generator; the device's log / exp round differently from libm's in a handful of draws per billion, and
 * the head tokens of a Zipf law are drawn in several independent parts per chunk, so the corpus has the same distribution
 * but is not bit for bit the host generator's).
 * vbm25_device_segment_download: the host copy (a vbm25_segment like any other: byte-identical to what the host builder
 * makes of the same mappings).  _token_terms / _query_bytes: as vbm25_segment_synth_token_terms / vbm25_query_bytes.
 * ---------------------------------------------------------------------- */
typedef struct vbm25_device_segment vbm25_device_segment;
int vbm25_device_segment_build(int device, double k1, double b, uint32_t n_docs, const uint32_t *doc_len,
                               const uint16_t *doc_payload, uint32_t n_terms, const uint8_t *term_key,
                               const uint64_t *term_start, const uint32_t *post_doc, const uint32_t *post_tf,
                               vbm25_device_segment **out);
int vbm25_device_segment_synth(const vbm25_synth_params *params, int device, vbm25_device_segment **out);
int vbm25_device_segment_download(const vbm25_device_segment *, vbm25_segment **out);
int vbm25_device_segment_token_terms(const vbm25_device_segment *, const uint32_t *tokens, uint32_t n, uint32_t *term_ids);
uint64_t vbm25_device_segment_query_bytes(const vbm25_device_segment *, const uint32_t *term_ids, uint32_t n_terms, uint32_t k);
int vbm25_device_segment_info(const vbm25_device_segment *, uint32_t *n_docs, uint32_t *n_terms, uint32_t *n_blocks,
                              uint64_t *n_postings);
void vbm25_device_segment_free(vbm25_device_segment *);
/* The index of a device segment, on the segment's device; the segment is left as it was. */
int vbm25_index_create_from_device(const vbm25_device_segment *, vbm25_index **out);

/* ------------------------------------------------------------------------
 * Several GPUs of one node: independent queries shard
 * across the devices, the index is replicated.  vbm25_multi_create uploads the
 * flattened segment to devices[0] ONCE and makes the other replicas GPU to GPU
 * (hipMemcpyPeerAsync over xGMI, derived arrays included; a device may be
 * listed more than once).  A batch is cut into contiguous, balanced shards --
 * the first nq % n devices get one query more --, every device searches its
 * shard on its own stream, and the 24-byte hit records go from every device
 * straight into the caller's host arrays in query order (the caller is the
 * host: a device-side gather would only add a hop).  Same results, record for
 * record, as vbm25_search_batch on one device.  Every device has its own host
 * thread inside the library (they belong to the vbm25_multi); a handle may be
 * used from one thread at a time, and calls on DIFFERENT vbm25_multi_batch
 * objects of one vbm25_multi from several threads are serialised by the library.
 * ---------------------------------------------------------------------- */
typedef struct vbm25_multi vbm25_multi;
typedef struct vbm25_multi_batch vbm25_multi_batch;
int vbm25_multi_create(const vbm25_index_desc *desc, const int *devices, int n_devices,
                       vbm25_multi **out);
void vbm25_multi_destroy(vbm25_multi *);
int vbm25_multi_device_count(const vbm25_multi *);
/* The replica on devices[i] (borrowed: e.g. for vbm25_lookup_terms, which is the same on every replica). */
int vbm25_multi_index(vbm25_multi *, int i, vbm25_index **out);
/* bm25::search for nq queries, sharded; arguments and results as vbm25_search_batch.  Synchronous. */
int vbm25_multi_search_batch(vbm25_multi *, const uint32_t *term_ids, const uint32_t *q_off,
                             uint32_t nq, uint32_t k, vbm25_hit *hits, uint32_t *n_hits);
/* The same with the shards resident on their devices: create once, set the queries, run (asynchronous:
 * every device's scan and the download of its records are enqueued on that device's stream), fetch (waits for
 * all devices and fills the caller's arrays).  run may be repeated. */
int vbm25_multi_batch_create(vbm25_multi *, uint32_t max_queries, uint32_t max_total_terms, uint32_t k,
                             vbm25_multi_batch **out);
void vbm25_multi_batch_destroy(vbm25_multi_batch *);
int vbm25_multi_batch_set_queries(vbm25_multi_batch *, const uint32_t *term_ids, const uint32_t *q_off,
                                  uint32_t nq);
int vbm25_multi_batch_run(vbm25_multi_batch *);
int vbm25_multi_batch_fetch(vbm25_multi_batch *, vbm25_hit *hits, uint32_t *n_hits);

#ifdef __cplusplus
}
#endif
#endif
