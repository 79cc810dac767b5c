/*
 * icrec.h — C ABI of libicrec.so, the MI355X (gfx950) implementation of the
 * reference's SBERT-encode -> cosine-similarity -> top-k hot path.
 *
 * The reference (chen-bowen/instacart_next_order_recommendation) has no FFI of
 * its own: its seam is the Python duck type `Recommender.recommend()`
 * (src/inference/serve_recommendations.py:206-225).  Each entry point below
 * names the reference call it replaces.  The Python host classes in
 * instacart_next_order_recommendation_amd/ bind these symbols with ctypes
 * (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *  - every function returns 0 on success, a negative ICREC_E* code on failure;
 *    icrec_last_error() returns a thread-local message for the last failure.
 *  - handles are opaque, owned by the library until the matching *_destroy.
 *  - "dev" pointers are device (HBM) pointers, borrowed for the duration of
 *    the call's stream work; "host" pointers are read before the call returns.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *    work is enqueued asynchronously; nothing here synchronises the device
 *    except *_create/_destroy.
 *  - scratch memory is supplied by the caller (`workspace`), sized by the
 *    matching *_workspace_bytes(); the library never allocates on the hot path
 *    so every call is hipGraph-capturable.
 *  - no torch / C++ types cross this boundary.
 */
#ifndef ICREC_H
#define ICREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* every entry point is exported from libicrec.so (built with -fvisibility=hidden) */
#define ICREC_API __attribute__((visibility("default")))

#define ICREC_VERSION_MAJOR 0
#define ICREC_VERSION_MINOR 1

enum {
    ICREC_OK = 0,
    ICREC_EINVAL = -1,   /* bad argument (shape, NULL, k out of range ...) */
    ICREC_EHIP = -2,     /* a HIP runtime call failed                      */
    ICREC_ENOMEM = -3,   /* workspace too small / allocation failed        */
    ICREC_ENODEV = -4    /* no gfx950 device visible                        */
};

/* k is bounded by the reference's API schema (src/api/schemas.py:34, top_k<=100);
 * the kernels are built for k <= ICREC_MAX_K. */
#define ICREC_MAX_K 128

typedef struct icrec_encoder icrec_encoder;
typedef struct icrec_index icrec_index;

/* ------------------------------------------------------------------------- */
/* Encoder: replaces SentenceTransformer.encode's device work                 */
/* (serve_recommendations.py:195-200, :213, :246): BertModel forward          */
/* (transformers modeling_bert.py BertEmbeddings/BertLayer), mean pooling,    */
/* L2 normalisation.  Tokenisation stays on the host.                         */
/* ------------------------------------------------------------------------- */

typedef struct icrec_bert_cfg {
    int32_t vocab_size;    /* 30522 for all-MiniLM-L6-v2                     */
    int32_t hidden;        /* 384  (must be 384 in this build)               */
    int32_t layers;        /* 6                                              */
    int32_t heads;         /* 12   (head_dim must be 32 in this build)       */
    int32_t intermediate;  /* 1536 (must be a multiple of 128)               */
    int32_t max_position;  /* 512                                            */
    int32_t type_vocab;    /* 2                                              */
    float   ln_eps;        /* 1e-12                                          */
    int32_t n_normalize;   /* how many times x / max(|x|_2, 1e-12) is applied
                              after pooling: 1 for the ST `Normalize` module,
                              +1 for encode(normalize_embeddings=True)       */
    int32_t gemm_mode;     /* ICREC_GEMM_F32: every linear layer on the exact
                              f32 MFMA (bit-identical to the oracle's fmaf
                              chains); ICREC_GEMM_F16X3: linear layers on the
                              f16 MFMA with 3-term operand splitting (fp32-level
                              accuracy, ~2^-21 relative per product; csrc/gemm_x3.h) */
} icrec_bert_cfg;

#define ICREC_GEMM_F32 0
#define ICREC_GEMM_F16X3 1

/* Number of fp32 elements the weight blob must hold for `cfg`.
 * Blob layout (all fp32, row-major, HF `nn.Linear` weights are [out,in]):
 *   word_emb[V,H] pos_emb[P,H] type_emb[Tv,H] emb_ln_g[H] emb_ln_b[H]
 *   then per layer:
 *   Wq[H,H] bq[H] Wk[H,H] bk[H] Wv[H,H] bv[H] Wo[H,H] bo[H] ln1_g[H] ln1_b[H]
 *   W1[I,H] b1[I] W2[H,I] b2[H] ln2_g[H] ln2_b[H]
 * (the BertPooler is not used by mean pooling and is not part of the blob). */
ICREC_API size_t icrec_encoder_weight_count(const icrec_bert_cfg* cfg);

/* Upload weights (host pointer; the library copies them to `device`). */
ICREC_API int icrec_encoder_create(const float* weights_host, size_t n_floats,
                         const icrec_bert_cfg* cfg, int device,
                         icrec_encoder** out);
ICREC_API int icrec_encoder_destroy(icrec_encoder* enc);

/* Scratch bytes needed to encode `total_tokens` tokens in `n_seqs` sequences. */
ICREC_API size_t icrec_encode_workspace_bytes(const icrec_encoder* enc,
                                    int64_t total_tokens, int32_t n_seqs);

/* Encode a token-packed batch.
 *   ids_dev        int32[total_tokens]  WordPiece ids, sequences back to back
 *                                       (already truncated to max_seq_length,
 *                                       [CLS]/[SEP] included; no pad tokens)
 *   cu_seqlens_dev int32[n_seqs+1]      prefix sums of sequence lengths
 *   max_seqlen     longest sequence in the batch (<= 256 in this build)
 *   out_dev        float[n_seqs, hidden] L2-normalised sentence embeddings
 * Padding never enters the math: the reference pads per batch and masks the
 * pad keys to weight exactly 0, so the packed form is the same function.
 * Stream semantics: asynchronous on `stream`; everything the call enqueues is ordered before whatever the caller
 * enqueues on `stream` afterwards.  For large batches (f16x3 mode) part of the work - the short attention buckets, the
 * remainder of icrec_encode_batch_split - runs on a library-owned side stream (one per caller stream, created on
 * first use) that forks from and joins back into `stream` by events inside the call; under stream capture the call
 * stays on `stream` unless that side stream already exists.  Calls on DIFFERENT streams may run concurrently (own
 * workspaces); calls on one encoder must not be issued from several host threads at once. */
ICREC_API int icrec_encode(icrec_encoder* enc,
                 const int32_t* ids_dev, const int32_t* cu_seqlens_dev,
                 int32_t n_seqs, int64_t total_tokens, int32_t max_seqlen,
                 float* out_dev,
                 void* workspace_dev, size_t workspace_bytes, void* stream);

/* How icrec_encode will split `total_tokens` (f16x3 mode): main_tokens go through the batch kernels (whole
 * rounds of one 64-token workgroup per CU; the fused FFN kernel sees exactly this many tokens), tail_tokens — a
 * remainder of at most 512 tokens — through the small-batch kernels.  Same arithmetic either way; bench.py uses
 * it to count the FLOPs of the launches it times. */
ICREC_API int icrec_encode_batch_split(const icrec_encoder* enc, int64_t total_tokens,
                             int64_t* main_tokens, int64_t* tail_tokens);

/* ------------------------------------------------------------------------- */
/* Index + search: replaces cos_sim(query_emb, product_embeddings)            */
/* (serve_recommendations.py:214/:250), scores.argsort(descending=True)       */
/* (:215/:251) and the exclusion/top-k loop (:216-225/:254-262).              */
/* ------------------------------------------------------------------------- */

/* Build an index over a [n_rows, dim] fp32 row-major matrix in device memory.
 * The library keeps its own copy with every row divided by max(|row|_2,1e-12)
 * (what cos_sim does to its second operand on every call in the reference).
 * `row_offset` is added to every returned row index (catalog shards).        */
ICREC_API int icrec_index_create(const float* rows_dev, int64_t n_rows, int32_t dim,
                       int64_t row_offset, int device, icrec_index** out);

/* Same, choosing how the normalised rows are kept in HBM (BASELINE config 5: a 10M x 384
 * bf16 catalog, 7.68 GB instead of 15.4 GB):
 *   ICREC_ROWS_F32   the fp32 quotient itself (what icrec_index_create does);
 *   ICREC_ROWS_BF16  the fp32 quotient rounded to bfloat16, round-to-nearest-even.
 * Only the storage changes: scores are still the fp32 fmaf chain over k of
 * q_hat[k] * float(row[k]) on the exact-f32 MFMA, so results are bit-identical to
 * oracle/icrec_oracle.c:icrec_oracle_search_bf16 (same rounded rows, same chain).       */
#define ICREC_ROWS_F32 0
#define ICREC_ROWS_BF16 1
/*   ICREC_ROWS_F32_FILTER  fp32 rows PLUS their f16 hi/lo planes (2x the HBM).  Batches of >= 256 queries
 *                    are first ranked on the f16 matrix cores (3 MFMAs per product, scores within ~1e-7, 5x
 *                    the fp32-MFMA rate) keeping k+12 candidates per query; every candidate is then re-scored
 *                    with the exact fp32 chain and the best k returned.  A query whose (k+12)-th candidate
 *                    is not provably below its exact k-th score (1e-4 margin) makes the exact search run
 *                    for the batch instead — results are therefore ALWAYS bit-identical to
 *                    ICREC_ROWS_F32, only faster for large catalogs x large batches.                    */
#define ICREC_ROWS_F32_FILTER 2
/*   ICREC_ROWS_BF16_FILTER  the same two-pass search over ICREC_ROWS_BF16 rows: bf16 rows (the exact pass and the
 *                    verification read these) plus f16 hi/lo planes of the rounded rows (3x the bf16 bytes
 *                    in total); bit-identical to ICREC_ROWS_BF16.                                        */
#define ICREC_ROWS_BF16_FILTER 3
ICREC_API int icrec_index_create_ex(const float* rows_dev, int64_t n_rows, int32_t dim,
                          int64_t row_offset, int device, int32_t storage, icrec_index** out);
ICREC_API int icrec_index_destroy(icrec_index* idx);
ICREC_API int64_t icrec_index_rows(const icrec_index* idx);
ICREC_API int64_t icrec_index_row_offset(const icrec_index* idx); /* global number of the shard's first row */
ICREC_API int32_t icrec_index_storage(const icrec_index* idx); /* ICREC_ROWS_* (-1: NULL handle) */
ICREC_API int32_t icrec_index_dim(const icrec_index* idx);     /* embedding width (0: NULL handle)  */
ICREC_API int32_t icrec_index_device(const icrec_index* idx);  /* HIP device ordinal (-1: NULL)     */

/* Copy the normalised rows back out (row-major fp32 [n_rows, dim]; bf16 storage is
 * widened exactly); used by the parity tests and by EmbeddingIndex.save.     */
ICREC_API int icrec_index_export(const icrec_index* idx, float* rows_dev, void* stream);

ICREC_API size_t icrec_search_workspace_bytes(const icrec_index* idx, int32_t n_queries,
                                    int32_t k);

/* Top-k search.
 *   q_dev        float[n_queries, dim]  query embeddings (any norm; normalised
 *                                       here exactly as cos_sim does)
 *   excl_idx_dev int32[excl_off[n_queries]] LOCAL row numbers to skip, sorted
 *                ascending and unique within each query's segment (or NULL)
 *   excl_off_dev int32[n_queries+1]     CSR offsets into excl_idx_dev (or NULL)
 *   out_idx_dev  int64[n_queries, k]    row_offset + row, best first; -1 pads
 *                                       when fewer than k rows remain
 *   out_score_dev float[n_queries, k]   cosine scores (0 where idx == -1)
 * Order: score descending, ties by lower row index first.
 * Every score is the fp32 chain s = fmaf(q[j], p[j], s) for j = 0..dim-1,
 * bit-identical to oracle/icrec_oracle.c:icrec_oracle_scores.                */
ICREC_API int icrec_search(icrec_index* idx, const float* q_dev, int32_t n_queries,
                 int32_t k,
                 const int32_t* excl_idx_dev, const int32_t* excl_off_dev,
                 int64_t* out_idx_dev, float* out_score_dev,
                 void* workspace_dev, size_t workspace_bytes, void* stream);

/* Shard-local half of a sharded search: same as icrec_search but emits the
 * sorted partial lists as packed 64-bit keys
 *   key = (orderable(score) << 32) | (0xFFFFFFFF - global_row)
 * so that a larger key is a better hit under (score desc, row asc).
 *   out_keys_dev uint64[n_queries, k], best first; 0 pads.                   */
ICREC_API int icrec_search_partial(icrec_index* idx, const float* q_dev,
                         int32_t n_queries, int32_t k,
                         const int32_t* excl_idx_dev,
                         const int32_t* excl_off_dev,
                         uint64_t* out_keys_dev,
                         void* workspace_dev, size_t workspace_bytes,
                         void* stream);

/* Merge `n_lists` sorted partial lists per query (e.g. the all-gathered
 * per-shard lists, laid out [n_lists, n_queries, k] as an all-gather leaves
 * them) into the final top-k.  Needs no index handle.                        */
ICREC_API int icrec_merge_topk(const uint64_t* keys_dev, int32_t n_lists,
                     int32_t n_queries, int32_t k,
                     int64_t* out_idx_dev, float* out_score_dev,
                     int device, void* stream);

/* Complete ranking of the catalog for each query: what `scores.argsort(descending=True)` returns in the reference's
 * offline evaluation consumers (src/baselines/content_based.py:58-63, scripts/compare_untrained_vs_trained.py:74-85).
 *   out_rows_dev int64[n_queries, n_rows]  row_offset + row, best first: score descending, lower row first on ties
 * (torch.argsort is unstable on ties; this is the library's total order, the same as icrec_search's).  The exact
 * score rows are materialised in the workspace (n_queries * n_rows * 12..20 bytes): callers stream queries in
 * passes (256 queries over 49,688 rows = 0.2 GB).  Not on the serving path.                                       */
ICREC_API size_t icrec_rank_all_workspace_bytes(const icrec_index* idx, int32_t n_queries);
ICREC_API int icrec_rank_all(icrec_index* idx, const float* q_dev, int32_t n_queries, int64_t* out_rows_dev,
                   void* workspace_dev, size_t workspace_bytes, void* stream);

/* Full score row(s) for parity checks: out[n_queries, n_rows] = q_hat . p_hat.
 * Not on the serving path (the serving kernels never materialise scores).    */
ICREC_API int icrec_scores(icrec_index* idx, const float* q_dev, int32_t n_queries,
                 float* out_dev, void* workspace_dev, size_t workspace_bytes,
                 void* stream);

/* L2-normalise rows in place-compatible fashion: out = x / max(|x|_2, eps).
 * (torch.nn.functional.normalize(p=2, dim=1) as used by cos_sim.)            */
ICREC_API int icrec_normalize_rows(const float* x_dev, float* out_dev, int64_t n_rows,
                         int32_t dim, float eps, int device, void* stream);

/* ------------------------------------------------------------------------- */
/* Multi-GPU exchange (new design, the reference has none:      */
/* serve_recommendations.py:172-181 only picks a device).  One process per    */
/* GPU; the catalog is row-sharded (each rank's icrec_index carries its       */
/* row_offset), queries are data-parallel.  The collectives are RCCL          */
/* all-gathers over xGMI, issued on the caller's stream from inside the       */
/* library (librccl.so.1 is bound with dlopen on first use).                  */
/* ------------------------------------------------------------------------- */
typedef struct icrec_comm icrec_comm;
#define ICREC_COMM_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

/* Rank 0 creates the rendezvous id (ncclGetUniqueId) into id_out[ICREC_COMM_ID_BYTES] (host memory) and
 * hands the bytes to every other rank by any out-of-band channel (file, socket, MPI, torch store). */
ICREC_API int icrec_comm_unique_id(void* id_out);
/* Collective over all `world` ranks: ncclCommInitRank on `device`.  world == 1 with unique_id == NULL makes a
 * communicator that exchanges nothing (icrec_search_sharded then equals icrec_search). */
ICREC_API int icrec_comm_init(const void* unique_id, int rank, int world, int device, icrec_comm** out);
ICREC_API int icrec_comm_destroy(icrec_comm* comm);
ICREC_API int32_t icrec_comm_rank(const icrec_comm* comm);
ICREC_API int32_t icrec_comm_world(const icrec_comm* comm);

ICREC_API size_t icrec_search_sharded_workspace_bytes(const icrec_index* idx, const icrec_comm* comm,
                                            int32_t n_local_queries, int32_t k);
/* Collective sharded top-k: every rank passes ITS n_local query embeddings (the same n_local on every rank)
 * and gets the global result for all Q = world * n_local queries, in rank-major order:
 *     ncclAllGather(q_local) -> icrec_search_partial over this rank's shard ->
 *     ncclAllGather(partial keys) -> icrec_merge_topk
 *   q_local_dev   float[n_local, dim]
 *   excl_idx_dev / excl_off_dev  CSR of LOCAL row numbers (this shard's rows) to skip for each of the Q
 *                                gathered queries (int32[Q+1] offsets), or NULL
 *   out_idx_dev   int64[Q, k] GLOBAL rows (row_offset + local row), -1 pads;  out_score_dev float[Q, k]
 * The result is bit-identical on every rank and to icrec_search over the unsharded catalog: the union of the
 * per-shard top-k lists contains the global top-k and the (score desc, row asc) order is total. */
ICREC_API int icrec_search_sharded(icrec_index* idx, icrec_comm* comm, const float* q_local_dev,
                         int32_t n_local_queries, int32_t k,
                         const int32_t* excl_idx_dev, const int32_t* excl_off_dev,
                         int64_t* out_idx_dev, float* out_score_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same collective with PER-RANK exclusion lists: the reference takes `exclude_product_ids` per request
 * (serve_recommendations.py:216-225), and with data-parallel front-ends only the rank that received a request
 * knows its list.  Every rank passes the exclusions of ITS n_local queries as GLOBAL row numbers; the library
 * all-gathers them (offsets, then the ids padded to excl_cap) and each rank applies the ones inside its shard:
 *     ncclAllGather(excl_off) + ncclAllGather(excl_rows)  -> shard-local CSR for the Q gathered queries (3 tiny kernels)
 *     then as icrec_search_sharded
 *   excl_rows_dev  int32[excl_cap]  GLOBAL rows, the CSR values of this rank's n_local queries: each query's rows
 *                                   ascending and unique; entries past excl_off[n_local] are ignored
 *   excl_off_dev   int32[n_local+1] offsets into excl_rows_dev (excl_off[n_local] <= excl_cap)
 *   excl_cap       the padded length of every rank's id buffer: the SAME on every rank (a deployment constant, e.g.
 *                  n_local x the API's per-request limit); 8 * world * excl_cap bytes of workspace
 * Offsets are sanitised on the device after the exchange (they arrive from other ranks): clamped to [0, excl_cap] and
 * made non-decreasing by a running maximum, so a rank's segments are disjoint and hold at most excl_cap ids (a malformed
 * list excludes less, never reads or writes out of bounds).
 * Row numbers travel as int32: catalogs of up to 2^31 - 1 rows for THIS exchange (the searches themselves take
 * row offsets to 4 * 10^9; the reference's catalog has 49,688 rows, BASELINE configs[4] 10^7). */
ICREC_API size_t icrec_search_sharded_excl_workspace_bytes(const icrec_index* idx, const icrec_comm* comm,
                                                 int32_t n_local_queries, int32_t k, int32_t excl_cap);
ICREC_API int icrec_search_sharded_excl(icrec_index* idx, icrec_comm* comm, const float* q_local_dev,
                              int32_t n_local_queries, int32_t k,
                              const int32_t* excl_rows_dev, const int32_t* excl_off_dev, int32_t excl_cap,
                              int64_t* out_idx_dev, float* out_score_dev,
                              void* workspace_dev, size_t workspace_bytes, void* stream);

/* The device-side step of that exchange on its own, for callers that move the lists with a transport of their own
 * (torch.distributed, MPI) and for testing the gathered layout without a second GPU: the rank-major buffers exactly
 * as ncclAllGather lays them down -> the CSR of LOCAL rows icrec_search / icrec_search_partial take for the shard
 * [row_lo, row_hi).
 *   off_all_dev   int32[world][n_local+1]   per rank: offsets of its n_local queries into its id buffer (not modified;
 *                                           sanitised in a workspace copy as described above)
 *   rows_all_dev  int32[world][excl_cap]    per rank: GLOBAL rows, each query's ascending and unique
 *   csr_off_dev   int32[world*n_local + 1]  out: offsets for the gathered queries (rank-major)
 *   csr_idx_dev   int32[world*excl_cap]     out: rows - row_lo of the ids inside the shard, order kept
 * Workspace: icrec_exclusions_to_shard_csr_workspace_bytes(world, n_local) bytes on `device`. */
ICREC_API size_t icrec_exclusions_to_shard_csr_workspace_bytes(int32_t world, int32_t n_local_queries);
ICREC_API int icrec_exclusions_to_shard_csr(const int32_t* off_all_dev, const int32_t* rows_all_dev, int32_t world,
                                  int32_t n_local_queries, int32_t excl_cap, int64_t row_lo, int64_t row_hi,
                                  int32_t* csr_off_dev, int32_t* csr_idx_dev,
                                  void* workspace_dev, size_t workspace_bytes, int device, void* stream);

/* ------------------------------------------------------------------------- */
/* Host tokenizer: the WordPiece stage of SentenceTransformer.encode           */
/* (serve_recommendations.py:213,:246 -> transformers BertTokenizer ->         */
/* tokenizers 0.22.2).  Pure host code; produces icrec_encode's packed input.  */
/* ------------------------------------------------------------------------- */
typedef struct icrec_tokenizer icrec_tokenizer;

/* vocab_path: BERT vocab.txt (one token per line, id = line number).
 * do_lower_case: lower-case + strip accents (all-MiniLM-L6-v2: 1).
 * max_len: [CLS] + tokens + [SEP] is truncated to this many ids (256). */
ICREC_API int icrec_tokenizer_create(const char* vocab_path, int do_lower_case, int max_len,
                           icrec_tokenizer** out);
ICREC_API int icrec_tokenizer_destroy(icrec_tokenizer* tok);
ICREC_API int32_t icrec_tokenizer_vocab_size(const icrec_tokenizer* tok);

/* Tokenise n UTF-8, NUL-terminated strings on up to n_threads host threads
 * (<= 0: all cores).  out_cu[n+1] receives the prefix sums of the id counts and
 * is always filled; the ids go to out_ids back to back.  Returns ICREC_ENOMEM
 * when `cap` ids do not suffice (size the buffer from out_cu[n] and retry). */
ICREC_API int icrec_tokenize(const icrec_tokenizer* tok, const char* const* texts, int32_t n,
                   int32_t* out_ids, int64_t cap, int32_t* out_cu, int32_t n_threads);

/* ------------------------------------------------------------------------- */
/* Diagnostics                                                                */
/* ------------------------------------------------------------------------- */
ICREC_API const char* icrec_last_error(void);
ICREC_API const char* icrec_version(void);
/* Average duration (ms) of the dominant kernel over the launches recorded
 * since the last reset, measured with hipEvents on the launch stream.
 * which: 0 = search score+select kernel, 1 = encoder FFN-up GEMM,
 *        2 = whole encode() call, 3 = whole search() call,
 *        4 = the guarded exact pass behind a filter pass (ICREC_ROWS_F32_FILTER):
 *            a few microseconds when every query was proven, a full search
 *            when the fallback ran.                                           */
ICREC_API int icrec_timing_enable(int on);
ICREC_API int icrec_timing_reset(void);
ICREC_API int icrec_timing_query(int which, double* avg_ms, int64_t* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* ICREC_H */
