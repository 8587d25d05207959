/* mnmf.h -- C-ABI of the MI355X-native NMF multiplicative-update engine.
 *
 * Drop-in boundary for MANTA's solver hot path.  The reference has no FFI: its seam is the
 * Python function run_nmf (manta/_functions/nmf/nmf_orchestrator.py:143).  Each entry point
 * below names the reference interface it replaces; manta_amd/nmf.py is the Python binding
 * with the reference signature (INTEGRATION.md shows the ctypes stub).
 *
 * Conventions
 *  - plain pointers and sizes only; no exceptions cross the boundary.
 *  - every function returns MNMF_OK (0) or a negative mnmf_status; mnmf_last_error(ctx)
 *    returns the text for the last failure on that context.
 *  - "device pointer" = HIP device memory on the context's GPU, caller-owned.
 *  - factor layouts are the reference's: W is m x k row-major, H is k x V row-major,
 *    S is k x k row-major, element type = the context dtype (f32 or f64).
 *  - all work is enqueued on the context's stream; the run_* functions return after the
 *    stream has drained (blocking, like the reference).
 *  - one host thread per context; a context is bound to one GPU.
 */
#ifndef MNMF_H
#define MNMF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mnmf_ctx mnmf_ctx;

typedef enum {
  MNMF_OK = 0,
  MNMF_EINVAL = -1, /* bad argument / shape */
  MNMF_EHIP = -2,   /* HIP runtime error */
  MNMF_ECOMM = -3,  /* the all-reduce hook failed */
  MNMF_ENAN = -4,   /* non-finite convergence norm */
  MNMF_ESTATE = -5  /* call out of order (e.g. run before set_matrix) */
} mnmf_status;

enum { MNMF_F32 = 0, MNMF_F64 = 1 };

/* Solver options.  A negative value (or 0 for the int fields marked so) selects the
 * reference's hard-coded constant for the method:
 *   nmf : max_iter 10000 (-> <=10001 updates), eps 1e-10           nmf_basic.py:26-27,57
 *   pnmf: max_iter 1000, norm_thresh 0.005, eps 1e-10, zero 1e-10  nmf_projective_basic.py:37-50
 *   nmtf: max_iter 1000 (-> <=1001 updates), eps 1e-9              nmtf.py:128,204           */
typedef struct {
  int32_t max_iter;       /* <0: reference default */
  int32_t fixed_iters;    /* >0: apply exactly this many updates and skip the convergence test */
  int32_t check_every;    /* convergence test every n iterations; <=0 -> 1 (the reference tests every iteration).
                             Ignored when fixed_iters>0 unless force_check is set. */
  int32_t force_check;    /* with fixed_iters: still compute the norms every check_every iterations */
  int32_t finalize;       /* 1: apply the reference's post-loop thresholds / normalisation; 0: raw factors */
  int32_t profile_phases; /* 1: bracket every kernel class with HIP events and fill mnmf_stats.phase_ms */
  double norm_thresh;     /* <0: 0.005 (what both pipelines pass, modeling_pipeline.py:65) */
  double eps;             /* <0: reference default */
  double zero_threshold;  /* <0: 1e-5 (run_nmf default, nmf_orchestrator.py:143); pnmf: 1e-10 */
} mnmf_opts;

typedef struct {
  int32_t iters;          /* updates applied to the returned factors */
  int32_t converged;      /* 1 if the reference's stopping rule fired before max_iter */
  double w_norm, h_norm, s_norm; /* last norms computed (spectral for nmf/pnmf, Frobenius for nmtf) */
  double ms_total;        /* wall time of the loop (host clock, stream drained) */
  double ms_kernels;      /* HIP-event time around the loop on the context's stream */
  double bytes_model;     /* algorithmic HBM bytes per iteration, SURVEY.md section 8(d) */
  double flops_model;     /* algorithmic flops per iteration */
  /* HIP-event time per kernel class, summed over the loop (profile_phases=1), in ms:
   * 0 Gram of W (m rows)   1 SpMM X^T.W   2 H-side products + update   3 Gram of H / k x k chains
   * 4 SpMM X.H^T           5 W-side update                6 convergence norms      7 unused */
  double phase_ms[8];
} mnmf_stats;

/* Sum-all-reduce hook for document-sharded runs (one process per GPU).  Called by the solvers
 * at their exchange points with a device pointer inside the exchange buffer, an element count
 * and the context dtype; must leave the summed values in place, ordered after prior work on
 * `stream` and before later work on it.  Return 0 on success.  No counterpart in the reference
 * (it is single-process); the Python side implements it with torch.distributed (RCCL). */
typedef int (*mnmf_allreduce_fn)(void* user, void* dev_ptr, int64_t count, int dtype, void* stream);

/* Collective hook with the three operations of the document-sharded schedule.  All of them
 * work IN PLACE on device memory of the engine, ordered after prior work on `stream` and before later work on it:
 *   MNMF_COMM_ALLREDUCE       dev_ptr[0 .. count)                      <- sum over ranks
 *   MNMF_COMM_REDUCE_SCATTER  dev_ptr = n_ranks blocks of `count` elements; block `rank` <- sum over ranks of that
 *                             block (the other blocks are left unspecified)
 *   MNMF_COMM_ALLGATHER       dev_ptr = n_ranks blocks of `count` elements; block r <- rank r's block r
 * Return 0 on success.  No counterpart in the reference; the Python side implements it with torch.distributed on
 * tensors that alias the engine's buffers (RCCL: reduce_scatter_tensor / all_gather_into_tensor, no staging copy). */
enum { MNMF_COMM_ALLREDUCE = 0, MNMF_COMM_REDUCE_SCATTER = 1, MNMF_COMM_ALLGATHER = 2 };
typedef int (*mnmf_comm_fn)(void* user, int op, void* dev_ptr, int64_t count, int dtype, void* stream);

/* -- lifetime ---------------------------------------------------------------------------- */
/* Create a context on GPU `device`, computing in `dtype`, enqueuing on `hip_stream`
 * (a hipStream_t; NULL = the default stream). */
int mnmf_create(mnmf_ctx** out, int device, int dtype, void* hip_stream);
void mnmf_destroy(mnmf_ctx* ctx);
const char* mnmf_last_error(const mnmf_ctx* ctx);
const char* mnmf_version(void);

/* Elements per row of a k-column factor in the engine's internal (zero padded) layout for this context's dtype: what
 * one nnz of X gathers.  Measurement helper (bench.py's gather-traffic bound); no counterpart in the reference. */
int mnmf_row_stride(const mnmf_ctx* ctx, int k);

/* Install (or clear, fn==NULL) the all-reduce hook.  `n_ranks` is informational (byte model). */
int mnmf_set_allreduce(mnmf_ctx* ctx, mnmf_allreduce_fn fn, void* user, int n_ranks);

/* Install (or clear, fn==NULL) the collective hook; `rank` in [0, n_ranks).  With it, standard NMF and NMTF exchange
 * the k x V partial by reduce-scatter, update only this rank's slice of H^T and all-gather the result (instead of
 * all-reduce + the whole update on every rank); Projective NMF, whose next product needs the summed matrix itself,
 * keeps the all-reduce. */
int mnmf_set_comm(mnmf_ctx* ctx, mnmf_comm_fn fn, void* user, int rank, int n_ranks);

/* -- the matrix -------------------------------------------------------------------------- */
/* Give the engine this rank's rows of X as CSR (replaces `in_mat.tocsr()` being handed to the
 * solvers, nmf_orchestrator.py:116).  rowptr: int64[m+1]; colidx: int32[nnz], ascending inside a
 * row; vals: dtype[nnz].  on_device!=0: the three arrays are device pointers, else host.
 * The engine copies them, builds X^T and its streaming layouts, and keeps everything resident
 * until the next set_matrix / destroy; the caller keeps ownership of its buffers. */
int mnmf_set_matrix(mnmf_ctx* ctx, int64_t m, int64_t V, int64_t nnz, const int64_t* rowptr,
                    const int32_t* colidx, const void* vals, int on_device);

/* -- solvers (device pointers, reference layouts, updated in place) ------------------------ */
/* Loop A, replaces _basic_nmf (nmf_basic.py:6-93).  W: m x k in/out, H: k x V in/out. */
int mnmf_run_nmf(mnmf_ctx* ctx, int k, void* W, void* H, const mnmf_opts* opts, mnmf_stats* stats);
/* Loop A for several topic counts at once: what OptimizationPipeline.run_optimization asks for when it calls run_nmf once
 * per candidate topic count on one matrix (optimization_pipeline.py:105-136, 193-198).  n problems with
 * ks[0] + ... + ks[n-1] <= 32 (fp32 contexts; the topic counts that share one padded 128-byte factor row) are iterated
 * together: one pass over X per SpMM serves all of them.  W[p]: m x ks[p], H[p]: ks[p] x V, device pointers, in/out;
 * stats: n entries (may be NULL).  Every problem stops by the reference's rule on its own norms and its result is
 * bit-identical to mnmf_run_nmf on that problem alone.  MNMF_EINVAL for fp64 contexts, sharded contexts and batches
 * whose topic counts add up to more than 32: run those one by one. */
int mnmf_run_nmf_sweep(mnmf_ctx* ctx, int n, const int32_t* ks, void* const* W, void* const* H, const mnmf_opts* opts,
                       mnmf_stats* stats);
/* Loop B, replaces projective_nmf (nmf_projective_basic.py:8-121).  W: in/out, H: out only
 * (= W^T X with the L2-column-normalised W when opts->finalize). */
int mnmf_run_pnmf(mnmf_ctx* ctx, int k, void* W, void* H, const mnmf_opts* opts, mnmf_stats* stats);
/* Loop C, replaces _core_nmtf_test (nmtf/nmtf.py:123-216).  W, S (k x k), H all in/out. */
int mnmf_run_nmtf(mnmf_ctx* ctx, int k, void* W, void* S, void* H, const mnmf_opts* opts,
                  mnmf_stats* stats);

/* -- kernel-level entry points (tests, benchmarks, profiling) ------------------------------ */
/* out = X . B (transpose==0; B: V x k, out: m x k) or X^T . B (transpose!=0; B: m x k, out: V x k);
 * the expressions `in_mat @ h.T` (nmf_basic.py:45) and `w.T @ in_mat` (nmf_basic.py:40, as its
 * transpose).  Row-major, leading dimension k. */
int mnmf_spmm(mnmf_ctx* ctx, int transpose, int k, const void* B, void* out);
/* out (k x k) = A^T . B over `rows` rows (B==NULL: A^T A): `w.T @ w` (nmf_basic.py:39),
 * `h @ h.T` (:44, on H^T), `W.T @ pay` (nmf_projective_basic.py:75). */
int mnmf_gram(mnmf_ctx* ctx, int64_t rows, int k, const void* A, const void* B, void* out);
/* out = A * (num / (A . G + eps))  (sqrt_mode: A * sqrt(num / (A . G + num . G2 + eps))):
 * the fused numerator/denominator/eps update of nmf_basic.py:41-42,46-47 and
 * nmf_projective_basic.py:76-79.  A, num, out: rows x k; G, G2: k x k. */
int mnmf_update(mnmf_ctx* ctx, int64_t rows, int k, const void* A, const void* num, const void* G,
                const void* G2, double eps, int sqrt_mode, void* out);
/* mnmf_gram for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  NO counterpart in the reference (nor is it a SURVEY.md section 8 row); same arguments and result as
 * mnmf_gram (out = A^T . B over `rows` rows, B == NULL: A^T A; A, B: rows x k row-major, out: k x k, device pointers,
 * context dtype).  It calls no hipMalloc / hipFree or allocating library routine and does not synchronise: two kernels
 * are enqueued on the context's stream and the call returns.  The partial Grams live in a static 8 MiB device array of
 * the library (one copy per GPU, held until the process ends), so k <= 256 (MNMF_EINVAL beyond, with a message) and
 * every call on a GPU shares that array: calls that overlap on one GPU (two streams, two contexts, two threads) race;
 * calls ordered on one stream do not.  Nothing in the array has to be zero on entry.  Summation order differs from
 * mnmf_gram's (fp32: vector ALUs here, matrix cores there): results agree to rounding, not bit for bit. */
int mnmf_gram_static(mnmf_ctx* ctx, int64_t rows, int k, const void* A, const void* B, void* out);
/* Diagnostic, no counterpart in the reference: how many device / pinned-host allocations (hipMalloc, hipHostMalloc)
 * the library has made in this process so far, over all contexts.  tests/test_zz_no_alloc.py reads it before and after
 * mnmf_gram_static. */
int64_t mnmf_alloc_count(void);
/* sigma_max(|A - B|) (norm_kind 0, `norm_func(np.abs(w - w_old), 2)`, nmf_basic.py:52) or its
 * Frobenius norm (norm_kind 1, nmtf.py:179) for rows x k matrices. */
int mnmf_diff_norm(mnmf_ctx* ctx, int64_t rows, int k, const void* A, const void* B, int norm_kind,
                   double* out);
/* The n largest entries of every row (by_column 0) or every column (by_column 1) of the device matrix A (rows x cols,
 * row-major, the context's dtype), in descending order; on equal values the larger index first.  This is the ranking
 * rule of topic_extractor.py:294-300 (`np.flip(np.argsort(H[i, :]))`, `W[:, i]`) evaluated where the factors live.
 * idx_out: groups x n int32 (device; -1 past the end), val_out: groups x n (device). */
int mnmf_topk(mnmf_ctx* ctx, int64_t rows, int64_t cols, const void* A, int by_column, int n, int32_t* idx_out,
              void* val_out);
/* NMTF topic-relationship summary, the computation inside s_matrix_confusion (nmtf/nmtf.py:62-80): for every column i
 * of the device matrix S (k x k row-major, context dtype, k <= 1024) the row of its maximum (np.argmax: the first one on
 * ties) and that value, ordered by value descending (`np.argsort(max_values)[::-1]`; equal values: larger column
 * first).  Entry r of the outputs (device, k each) describes the r-th strongest coupling: col_out[r] = the column i,
 * row_out[r] = argmax row, val_out[r] = S[row, i].  Drains the stream. */
int mnmf_s_confusion(mnmf_ctx* ctx, int k, const void* S, int32_t* col_out, int32_t* row_out, void* val_out);
/* -- TF-IDF construction (the producer of X) --------------------------- */
/* Term-count matrix of a tokenised corpus, replacing the Counter / lil_matrix loop of
 * tfidf_english_calculator.py:80-87 and tfidf_turkish_calculator.py:37-45.  doc_ptr: int64[m+1] (device), document d
 * owns tokens[doc_ptr[d] .. doc_ptr[d+1]) (int32 vocabulary ids, device).  Pass 1 sorts and run-length encodes every
 * document: row_nnz[d] = number of distinct tokens, and the distinct ids (ascending) with their counts are left in the
 * caller's workspace ws_cols / ws_counts (int32[n_tokens] each, device) at the document's own token offset.  The caller
 * turns row_nnz into rowptr (exclusive prefix sum, rowptr[m] = nnz) and pass 2 moves each row to its CSR position (the
 * canonical CSR of `matris.tocsr()`) and writes df[V] = number of documents containing each word (`np.add.reduceat`
 * over the binary CSC copy, :90-92 / :47-49, without its quirk for empty columns, which the Python side reproduces).
 * MNMF_EINVAL: a token outside [0, V) or a document of more than 8192 tokens (mnmf_last_error says which).
 * Both calls drain the stream. */
int mnmf_tfidf_count_terms(mnmf_ctx* ctx, int64_t m, int64_t V, const int64_t* doc_ptr, const int32_t* tokens,
                           int64_t* row_nnz, int32_t* ws_cols, int32_t* ws_counts);
int mnmf_tfidf_fill_terms(mnmf_ctx* ctx, int64_t m, int64_t V, const int64_t* doc_ptr, const int64_t* rowptr, int64_t nnz,
                          const int32_t* ws_cols, const int32_t* ws_counts, int32_t* colidx, int32_t* counts, int32_t* df);
/* vals[p] = idf[colidx[p]] / pivot[row] * weight[row] in IEEE double, one rounding per operation: the entry values of
 * `tf_l(binary).multiply(idf)` followed by the pivoted normalisation and the PageRank weighting
 * (tfidf_english_calculator.py:98-127, tfidf_turkish_calculator.py:55-77).  pivot / weight may be NULL (step absent).
 * All pointers are device pointers; asynchronous on the context's stream. */
int mnmf_tfidf_scale(mnmf_ctx* ctx, int64_t m, const int64_t* rowptr, const int32_t* colidx, const double* idf,
                     const double* pivot, const double* weight, double* vals);
/* Asynchronous device-to-device copy on the context's stream.  No counterpart in the reference;
 * lets a Python all-reduce hook stage an exchange buffer it only knows by address. */
int mnmf_memcpy_d2d(mnmf_ctx* ctx, void* dst, const void* src, int64_t bytes);

/* -- host-only helpers (no GPU needed; used by the CPU tests) ------------------------------ */
/* Build the streaming-segment description exactly as the engine does (DESIGN.md "stream layout").
 * A matrix is a sequence of PIECES (runs of nnz with one destination).  mnmf_build_segments takes a CSR
 * row pointer (one piece per non-empty row); mnmf_build_segments_pieces takes explicit pieces in
 * processing order (piece i = nnz [piece_start[i], piece_start[i+1]) of output row piece_row[i]), which
 * is how the document-blocked X^T is described.  Pieces longer than sub_piece are cut; rows made of
 * several (sub-)pieces are summed from carry slots.  Outputs are malloc'ed; free with mnmf_free_segments. */
typedef struct {
  int64_t n_seg;       /* segments */
  int64_t* seg_start;  /* [n_seg] offset of the segment's first nnz */
  int32_t* seg_len;    /* [n_seg] nnz in the segment (0: padding segment) */
  int32_t* seg_info;   /* [n_seg] index into dest of the segment's first sub-piece */
  int64_t n_dest;      /* sub-pieces */
  int32_t* dest;       /* [n_dest] >=0: output row; <0: ~carry slot */
  int64_t* piece_end;  /* [n_dest] position of the sub-piece's last nnz (carries the end flag) */
  int64_t n_empty;
  int32_t* empty_rows; /* rows without nnz */
  int64_t n_fix;       /* rows summed from carry slots */
  int32_t* fix_row;    /* [n_fix] row id */
  int32_t* fix_first;  /* [n_fix] first carry slot */
  int32_t* fix_count;  /* [n_fix] number of carry slots */
  int64_t n_carry;
} mnmf_segments;
int mnmf_build_segments(const int64_t* rowptr, int64_t rows, int64_t seg_nnz, int64_t sub_piece,
                        mnmf_segments* out);
/* piece_block (may be NULL): document block of each piece, -1 = not blocked.  With piece_block and
 * xcd_group > 0 the segments are emitted in XCD-affine order: workgroup g (xcd_group segments each)
 * takes the next segments of the blocks b with b mod 8 == g mod 8; lists are padded with empty segments. */
int mnmf_build_segments_pieces(const int64_t* piece_start, const int32_t* piece_row, const int32_t* piece_block,
                               int64_t n_pieces, int64_t rows, int64_t seg_nnz, int64_t sub_piece,
                               int xcd_group, mnmf_segments* out);
/* The same with exact-length segments (exact != 0): a segment is closed after exactly seg_nnz nnz -- the piece is cut
 * there and both parts get carry slots -- which is what the engine does for small matrices. */
int mnmf_build_segments_exact(const int64_t* piece_start, const int32_t* piece_row, const int32_t* piece_block,
                              int64_t n_pieces, int64_t rows, int64_t seg_nnz, int64_t sub_piece,
                              int xcd_group, int exact, mnmf_segments* out);
/* Tile stream of X^T (the most frequent words; DESIGN.md "tile stream"): piece_tile[i] = tile of piece i,
 * ascending.  seg_nnz > 0: segments of about seg_nnz nnz that never cross a tile; seg_nnz <= 0: every tile is cut
 * into -seg_nnz equal segments (pieces are cut at segment boundaries).  tile_first (n_tiles + 1 entries, caller
 * allocated) receives the first segment of each tile.  Rows without pieces are not reported as empty. */
int mnmf_build_segments_tiles(const int64_t* piece_start, const int32_t* piece_row, const int32_t* piece_tile,
                              int64_t n_pieces, int64_t rows, int64_t n_tiles, int64_t seg_nnz, int64_t sub_piece,
                              mnmf_segments* out, int32_t* tile_first);
void mnmf_free_segments(mnmf_segments* s);
/* lambda_max of a symmetric n x n matrix (row-major, double) by cyclic Jacobi. */
double mnmf_sym_lambda_max(const double* a, int n);

#ifdef __cplusplus
}
#endif
#endif /* MNMF_H */
