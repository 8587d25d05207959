// TF-IDF construction on the device: the step right before the solver path.
// The reference builds a document x vocabulary count matrix with a Python Counter per document and a scipy
// lil_matrix (manta/_functions/tfidf/tfidf_english_calculator.py:80-87, tfidf_turkish_calculator.py:37-45), takes the
// document frequencies from its binary CSC copy (:90-92 / :47-49) and scales every stored entry by idf(word) /
// pivoted_norm(document) [* pagerank(document)] (:98-127 / :55-77).  Here a document's tokens are sorted and
// run-length encoded inside LDS by one workgroup (bitonic network, no global sort, tokens read once per pass), which
// yields the CSR row of the count matrix in canonical (ascending column) order; document frequencies are an integer
// histogram over the CSR columns; the scaling is one pass over the entries with IEEE double division and
// multiplication, the same two roundings as the reference's numpy expressions.  All integer outputs are exact.
#include <algorithm>
#include <string>

#include "common.hpp"
#include "ctx.hpp"
#include "histogram.hpp"

using namespace mnmf;

namespace {

constexpr int kSmallDoc = 1024;  // tokens: one wave sorts the document
constexpr int kLargeDoc = 8192;  // tokens: four waves; longer documents are refused (the host falls back to a global sort)

// exclusive prefix sum of one int per thread over the workgroup; `total` = sum.  tmp: THREADS / 64 + 1 ints of LDS.
template <int THREADS>
__device__ inline int block_excl_scan(int c, int* tmp, int& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if constexpr (THREADS == 64) {
    total = __shfl(inc, 63, 64);
    return inc - c;
  } else {
    __syncthreads();  // tmp may still be read from the previous call
    if (lane == 63) tmp[wv] = inc;
    __syncthreads();
    int base = 0, t = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
      const int v = tmp[w];
      if (w < wv) base += v;
      t += v;
    }
    total = t;
    return base + inc - c;
  }
}

// One workgroup per document (grid-stride): row_nnz[d] = number of distinct tokens; the distinct ids (ascending) and
// their counts go to the workspace at the document's own token offset (ws_cols / ws_counts [doc_ptr[d] ..]), from where
// tfidf_compact_kernel moves them to their CSR position once rowptr is known -- the tokens are sorted once.
// Documents whose length is outside (lo_len, hi_len] are left to
// the launch of the other size class.  flags: bit 0 = a token outside [0, V), bit 1 = a document longer than kLargeDoc,
// bit 2 = the small-class launch met a document of the large class (the large-class launch is skipped otherwise).
template <int THREADS, int CAP>
__global__ __launch_bounds__(THREADS) void tfidf_doc_terms_kernel(const int64_t* __restrict__ doc_ptr,
                                                                  const int32_t* __restrict__ tokens, int64_t m, int64_t V,
                                                                  int lo_len, int hi_len, int64_t* __restrict__ row_nnz,
                                                                  int32_t* __restrict__ ws_cols, int32_t* __restrict__ ws_counts,
                                                                  int* __restrict__ flags) {
  __shared__ uint32_t s[CAP];
  __shared__ int tmp[THREADS / 64 + 1];
  const int tid = threadIdx.x;
  for (int64_t d = blockIdx.x; d < m; d += gridDim.x) {
    const int64_t base = doc_ptr[d];
    const int64_t len64 = doc_ptr[d + 1] - base;
    if (len64 > kLargeDoc) {
      if (tid == 0 && hi_len == kSmallDoc) {
        atomicOr(flags, 2);
        row_nnz[d] = 0;
      }
      continue;
    }
    const int len = (int)len64;
    if (len > hi_len) {                           // left to the launch of the larger size class
      if (tid == 0) atomicOr(flags, 4);
      continue;
    }
    if (len <= lo_len) continue;                  // workgroup-uniform
    int n = 2;
    while (n < len) n <<= 1;
    __syncthreads();  // the previous document's readers are done with s
    bool bad = false;
    for (int i = tid; i < n; i += THREADS) {
      uint32_t t = 0xFFFFFFFFu;  // padding sorts last
      if (i < len) {
        const int32_t tok = tokens[base + i];
        bad |= (tok < 0 || (int64_t)tok >= V);
        t = (uint32_t)tok;
      }
      s[i] = t;
    }
    if (bad) atomicOr(flags, 1);
    __syncthreads();
    for (int k = 2; k <= n; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < n; i += THREADS) {
          const int x = i ^ j;
          if (x > i) {
            const uint32_t a = s[i], b = s[x];
            const bool asc = (i & k) == 0;
            if ((a > b) == asc) {
              s[i] = b;
              s[x] = a;
            }
          }
        }
        __syncthreads();
      }
    }
    // heads of runs: thread t owns positions [t*E, t*E + E)
    const int E = n >= THREADS ? n / THREADS : 1;
    const int p0 = tid * E, p1 = (p0 + E < len) ? p0 + E : len;
    int c = 0;
    for (int i = p0; i < p1; ++i) c += (i == 0 || s[i] != s[i - 1]);
    int total;
    int r = block_excl_scan<THREADS>(c, tmp, total);
    if (tid == 0) row_nnz[d] = total;
    for (int i = p0; i < p1; ++i) {
      if (i == 0 || s[i] != s[i - 1]) {
        const uint32_t v = s[i];
        int e = i + 1;
        while (e < len && s[e] == v) ++e;  // run end (may reach into the next thread's positions)
        ws_cols[base + r] = (int32_t)v;
        ws_counts[base + r] = e - i;
        ++r;
      }
    }
  }
}

// Row d of the workspace (at doc_ptr[d], rowptr[d+1] - rowptr[d] entries) to its CSR position: 16 lanes per row.
__global__ void tfidf_compact_kernel(int64_t m, const int64_t* __restrict__ doc_ptr, const int64_t* __restrict__ rowptr,
                                     const int32_t* __restrict__ ws_cols, const int32_t* __restrict__ ws_counts,
                                     int32_t* __restrict__ colidx, int32_t* __restrict__ counts) {
  const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t row = g; row < m; row += stride) {
    const int64_t src = doc_ptr[row], dst = rowptr[row];
    const int64_t n = rowptr[row + 1] - dst;
    for (int64_t j = l; j < n; j += 16) {
      colidx[dst + j] = ws_cols[src + j];
      counts[dst + j] = ws_counts[src + j];
    }
  }
}

// vals[p] = idf[colidx[p]] / pivot[row] * weight[row]: 16 lanes per row.  Two separately rounded operations, as in
// `tf_idf.data / np.repeat(pivoted_norms, ...)` then `* np.repeat(pagerank_weights, ...)`.
__global__ void tfidf_scale_kernel(int64_t m, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                   const double* __restrict__ idf, const double* __restrict__ pivot,
                                   const double* __restrict__ weight, double* __restrict__ vals) {
  const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t row = g; row < m; row += stride) {
    const int64_t b = rowptr[row], e = rowptr[row + 1];
    const double pv = pivot ? pivot[row] : 1.0;
    const double wt = weight ? weight[row] : 1.0;
    for (int64_t p = b + l; p < e; p += 16) {
      double v = idf[colidx[p]];
      if (pivot) v = __ddiv_rn(v, pv);
      if (weight) v = __dmul_rn(v, wt);
      vals[p] = v;
    }
  }
}

int check_launch(mnmf_ctx* ctx, const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ctx->fail(MNMF_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    return MNMF_EHIP;
  }
  return 0;
}

int launch_doc_terms(mnmf_ctx* ctx, int64_t m, int64_t V, const int64_t* doc_ptr, const int32_t* tokens, int64_t* row_nnz,
                     int32_t* ws_cols, int32_t* ws_counts, int* flags) {
  const unsigned grid_small = (unsigned)std::min<int64_t>(m, 256 * 64);
  const unsigned grid_large = (unsigned)std::min<int64_t>(m, 256 * 8);
  tfidf_doc_terms_kernel<64, kSmallDoc><<<grid_small, 64, 0, ctx->stream>>>(doc_ptr, tokens, m, V, -1, kSmallDoc, row_nnz, ws_cols,
                                                                           ws_counts, flags);
  MNMF_TRY(check_launch(ctx, "tfidf_doc_terms_kernel (small)"));
  int h = 0;  // documents of the large class are rare: look before launching a second scan over all documents
  if (hipMemcpyAsync(&h, flags, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    ctx->fail(MNMF_EHIP, "tfidf_doc_terms_kernel (small) failed");
    return MNMF_EHIP;
  }
  if (!(h & 4)) return 0;
  tfidf_doc_terms_kernel<256, kLargeDoc><<<grid_large, 256, 0, ctx->stream>>>(doc_ptr, tokens, m, V, kSmallDoc, kLargeDoc, row_nnz,
                                                                             ws_cols, ws_counts, flags);
  return check_launch(ctx, "tfidf_doc_terms_kernel (large)");
}

}  // namespace

extern "C" {

int mnmf_tfidf_count_terms(mnmf_ctx* ctx, int64_t m, int64_t V, const int64_t* doc_ptr, const int32_t* tokens,
                           int64_t* row_nnz, int32_t* ws_cols, int32_t* ws_counts) {
  if (!ctx) return MNMF_EINVAL;
  if (m < 0 || V < 1 || V > INT32_MAX || !doc_ptr || !row_nnz) {
    ctx->fail(MNMF_EINVAL, "mnmf_tfidf_count_terms: bad arguments");
    return MNMF_EINVAL;
  }
  if (m == 0) return MNMF_OK;
  MNMF_HIP_TRY(ctx, hipSetDevice(ctx->device));
  int* flags = nullptr;
  MNMF_HIP_TRY(ctx, mnmf_counted_malloc((void**)&flags, sizeof(int)));
  int rc = MNMF_OK, h = 0;
  if (hipMemsetAsync(flags, 0, sizeof(int), ctx->stream) != hipSuccess) rc = MNMF_EHIP;
  if (rc == 0) rc = launch_doc_terms(ctx, m, V, doc_ptr, tokens, row_nnz, ws_cols, ws_counts, flags);
  if (rc == 0 && (hipMemcpyAsync(&h, flags, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                  hipStreamSynchronize(ctx->stream) != hipSuccess)) {
    ctx->fail(MNMF_EHIP, "mnmf_tfidf_count_terms: reading the flags failed");
    rc = MNMF_EHIP;
  }
  (void)hipFree(flags);
  if (rc != 0) return rc;
  if (h & 2) {
    ctx->fail(MNMF_EINVAL, "mnmf_tfidf_count_terms: a document has more than " + std::to_string(kLargeDoc) + " tokens");
    return MNMF_EINVAL;
  }
  if (h & 1) {
    ctx->fail(MNMF_EINVAL, "mnmf_tfidf_count_terms: token id outside [0, V)");
    return MNMF_EINVAL;
  }
  return MNMF_OK;
}

int mnmf_tfidf_fill_terms(mnmf_ctx* ctx, int64_t m, int64_t V, const int64_t* doc_ptr, const int64_t* rowptr, int64_t nnz,
                          const int32_t* ws_cols, const int32_t* ws_counts, int32_t* colidx, int32_t* counts, int32_t* df) {
  if (!ctx) return MNMF_EINVAL;
  if (m < 0 || V < 1 || V > INT32_MAX || nnz < 0 || !doc_ptr || !rowptr || !df ||
      (nnz > 0 && (!colidx || !counts || !ws_cols || !ws_counts))) {
    ctx->fail(MNMF_EINVAL, "mnmf_tfidf_fill_terms: bad arguments");
    return MNMF_EINVAL;
  }
  MNMF_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (m == 0 || nnz == 0) {
    MNMF_HIP_TRY(ctx, hipMemsetAsync(df, 0, (size_t)V * 4, ctx->stream));
    return MNMF_OK;
  }
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(m * 16, 256), 256 * 32);
  tfidf_compact_kernel<<<grid, 256, 0, ctx->stream>>>(m, doc_ptr, rowptr, ws_cols, ws_counts, colidx, counts);
  int rc = check_launch(ctx, "tfidf_compact_kernel");
  // document frequencies: histogram over the distinct (document, word) pairs (histogram.hpp)
  const int copies = hist_copies(V);
  int32_t* parts = nullptr;
  if (rc == 0 && mnmf_counted_malloc((void**)&parts, (size_t)copies * V * 4) != hipSuccess) {
    ctx->fail(MNMF_EHIP, "mnmf_tfidf_fill_terms: out of device memory");
    rc = MNMF_EHIP;
  }
  if (rc == 0 && launch_histogram((const uint32_t*)colidx, nnz, V, 0xFFFFFFFFu, parts, copies, df, ctx->stream) != hipSuccess) {
    ctx->fail(MNMF_EHIP, "mnmf_tfidf_fill_terms: histogram launch failed");
    rc = MNMF_EHIP;
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == 0) {
    ctx->fail(MNMF_EHIP, "mnmf_tfidf_fill_terms: kernel failed");
    rc = MNMF_EHIP;
  }
  if (parts) (void)hipFree(parts);
  return rc;
}

int mnmf_tfidf_scale(mnmf_ctx* ctx, int64_t m, const int64_t* rowptr, const int32_t* colidx, const double* idf,
                     const double* pivot, const double* weight, double* vals) {
  if (!ctx) return MNMF_EINVAL;
  if (m < 0 || !rowptr || !idf) {
    ctx->fail(MNMF_EINVAL, "mnmf_tfidf_scale: bad arguments");
    return MNMF_EINVAL;
  }
  if (m == 0) return MNMF_OK;
  MNMF_HIP_TRY(ctx, hipSetDevice(ctx->device));
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(m * 16, 256), 256 * 32);
  tfidf_scale_kernel<<<grid, 256, 0, ctx->stream>>>(m, rowptr, colidx, idf, pivot, weight, vals);
  return check_launch(ctx, "tfidf_scale_kernel");
}

}  // extern "C"
