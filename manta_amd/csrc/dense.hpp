// Dense kernels of the MU iteration: tall-skinny Gram products (K3), the rows x (k x k) products
// fused with the multiplicative update (K4+K5), the k x k chains of NMTF, layout conversion and
// finalisation (K7).  All factor matrices are rows x ld row-major with ld = round_up(k, 4) and zero
// padding; all k x k matrices are ld x ld zero-padded, so no kernel needs a column mask below ld.
#pragma once
#include "common.hpp"

namespace mnmf {

// ------------------------------------------------------------------------------------ Gram (VALU)
// slab[chunk] (ld x ld) = A_chunk^T . B_chunk over rows [chunk*rpc, (chunk+1)*rpc).
// DIFF: the operands are |A - A2| and |B - B2| (Gram of the update step for the spectral norm,
// nmf_basic.py:52-53 via sigma_max(D)^2 = lambda_max(D^T D)).
// grid = (n_chunks, tiles_i * tiles_j); block 256 = 16 x 16 threads, 4 x 4 outputs each.
template <typename T, bool DIFF>
__global__ __launch_bounds__(256) void gram_partial_kernel(const T* __restrict__ A, const T* __restrict__ A2,
                                                           const T* __restrict__ B, const T* __restrict__ B2,
                                                           int64_t rows, int ld, int64_t rpc,
                                                           T* __restrict__ slab) {
  constexpr int TR = 16;
  __shared__ T As[TR][64];
  __shared__ T Bs[TR][64];
  const int tiles = (ld + 63) / 64;
  const int ti = blockIdx.y / tiles, tj = blockIdx.y % tiles;
  const int ci0 = ti * 64, cj0 = tj * 64;
  const int64_t r_begin = (int64_t)blockIdx.x * rpc;
  const int64_t r_end = (r_begin + rpc < rows) ? r_begin + rpc : rows;
  const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) acc[i][jx] = T(0);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += TR) {
#pragma unroll
    for (int q = 0; q < (TR * 64) / 256; ++q) {
      const int idx = t + q * 256;
      const int rr = idx >> 6, cc = idx & 63;
      const int64_t row = r0 + rr;
      T a = T(0), b = T(0);
      if (row < r_end) {
        if (ci0 + cc < ld) {
          a = A[row * ld + ci0 + cc];
          if (DIFF) a = fabs(a - A2[row * ld + ci0 + cc]);
        }
        if (cj0 + cc < ld) {
          b = B[row * ld + cj0 + cc];
          if (DIFF) b = fabs(b - B2[row * ld + cj0 + cc]);
        }
      }
      As[rr][cc] = a;
      Bs[rr][cc] = b;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < TR; ++rr) {
      T a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[rr][ty * 4 + i];
#pragma unroll
      for (int jx = 0; jx < 4; ++jx) b[jx] = Bs[rr][tx * 4 + jx];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) acc[i][jx] += a[i] * b[jx];
    }
    __syncthreads();
  }
  T* dst = slab + (int64_t)blockIdx.x * ld * ld;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gi_ = ci0 + ty * 4 + i;
    if (gi_ >= ld) continue;
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      const int gj = cj0 + tx * 4 + jx;
      if (gj < ld) dst[(int64_t)gi_ * ld + gj] = acc[i][jx];
    }
  }
}

// out[e] = sum over chunks of slab[chunk][e], in a fixed order (deterministic): 16 threads per element
// take chunks c = t, t+16, ... each, then their 16 partial sums are added in lane order.
// block 256 = 16 elements x 16 chunk lanes.
template <typename T>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const T* __restrict__ slab, int64_t n_chunks,
                                                          int64_t elems, T* __restrict__ out,
                                                          const T* __restrict__ slab2 = nullptr, T* __restrict__ out2 = nullptr) {
  // blockIdx.y == 1: the second slab of a fused launch (two Grams from one update kernel)
  if (blockIdx.y == 1) {
    slab = slab2;
    out = out2;
  }
  __shared__ T part[16][17];
  const int el = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int64_t e = (int64_t)blockIdx.x * 16 + el;
  T s = T(0);
  if (e < elems)
    for (int64_t c = cl; c < n_chunks; c += 16) s += slab[c * elems + e];
  part[cl][el] = s;
  __syncthreads();
  if (cl == 0 && e < elems) {
    T t = T(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][el];
    out[e] = t;
  }
}

// -------------------------------------------------------------- rows x (k x k) products + MU update
enum RkMode { RK_PROD = 0, RK_MU = 1, RK_MU_SQRT = 2 };

template <typename T>
struct RkArgs {
  const T* A1;   // rows x ld
  const T* G1;   // ld x ld
  const T* A2;   // optional second pair (pnmf: pay . wtw), may be null
  const T* G2;
  const T* upd;  // matrix being updated (MU modes)
  const T* num;  // numerator (MU modes)
  T* out;        // rows x ld (never aliases A1/A2)
  int64_t rows;
  int ld;
  T eps;
  // Batched topic-count sweep (matrix-core kernel, ld <= 32 only): the factor buffers hold several problems side by
  // side in one padded row; this call works on the problem whose `kcols` columns start at column `coff`.  Logical
  // column c of the product / update is physical column coff + c of A1, upd, num and out; G1 is the problem's own
  // ld x ld matrix (zero beyond kcols).  kcols == 0: a plain call (all ld columns, coff 0).
  int coff = 0;
  int kcols = 0;
};

// acc = A1.G1 (+ A2.G2);  PROD: out = acc;  MU: out = upd * (num / (acc + eps));
// MU_SQRT: out = upd * sqrt(num / (acc + eps)).
// grid = (ceil(rows/64), ceil(ld/64)); block 256 = 16 x 16 threads, 4 x 4 outputs each.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void dense_rk_kernel(RkArgs<T> a) {
  constexpr int TK = 16;
  __shared__ T As[64][TK + 1];
  __shared__ T Gs[TK][64];
  const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int ld = a.ld;
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) acc[i][jx] = T(0);
  for (int pair = 0; pair < 2; ++pair) {
    const T* A = pair == 0 ? a.A1 : a.A2;
    const T* G = pair == 0 ? a.G1 : a.G2;
    if (A == nullptr) break;
    for (int k0 = 0; k0 < ld; k0 += TK) {
#pragma unroll
      for (int q = 0; q < (64 * TK) / 256; ++q) {
        const int idx = t + q * 256;
        const int rr = idx / TK, kk = idx % TK;
        const int64_t row = row0 + rr;
        As[rr][kk] = (row < a.rows && k0 + kk < ld) ? A[row * ld + k0 + kk] : T(0);
      }
#pragma unroll
      for (int q = 0; q < (TK * 64) / 256; ++q) {
        const int idx = t + q * 256;
        const int kk = idx >> 6, cc = idx & 63;
        Gs[kk][cc] = (k0 + kk < ld && c0 + cc < ld) ? G[(int64_t)(k0 + kk) * ld + c0 + cc] : T(0);
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < TK; ++kk) {
        T av[4], gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = As[ty * 4 + i][kk];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) gv[jx] = Gs[kk][tx * 4 + jx];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) acc[i][jx] += av[i] * gv[jx];
      }
      __syncthreads();
    }
  }
  const int col = c0 + tx * 4;
  if (col >= ld) return;  // ld % 4 == 0: a thread's 4 columns are all inside or all outside
  using V4 = Vec<T, 4>;
  // loads of all four rows before the first store: out may alias nothing here, but the compiler cannot know
  V4 u[4], n[4];
  if (MODE != RK_PROD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = row0 + ty * 4 + i;
      const int64_t off = (row < a.rows ? row : a.rows - 1) * ld + col;
      u[i] = *reinterpret_cast<const V4*>(a.upd + off);
      n[i] = *reinterpret_cast<const V4*>(a.num + off);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t row = row0 + ty * 4 + i;
    if (row >= a.rows) continue;
    const int64_t off = row * ld + col;
    V4 o;
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      if (MODE == RK_PROD) {
        o.v[jx] = acc[i][jx];
      } else {
        const T q = n[i].v[jx] / (acc[i][jx] + a.eps);
        o.v[jx] = u[i].v[jx] * (MODE == RK_MU_SQRT ? sqrt(q) : q);
      }
    }
    *reinterpret_cast<V4*>(a.out + off) = o;
  }
}

// ------------------------------------------------------------------------------- k x k helpers
// C = op(A) . op(B) on n x n row-major matrices (n = ld); thread per output element.
template <typename T>
__global__ void kxk_gemm_kernel(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C, int n,
                                int ta, int tb) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  const int i = idx / n, jx = idx % n;
  T s = T(0);
  for (int l = 0; l < n; ++l) {
    const T av = ta ? A[l * n + i] : A[i * n + l];
    const T bv = tb ? B[jx * n + l] : B[l * n + jx];
    s += av * bv;
  }
  C[idx] = s;
}

// out = s * (num / (den + eps)) elementwise on n x n (the S update of nmtf.py:156-158).
template <typename T>
__global__ void kxk_mu_kernel(const T* __restrict__ s, const T* __restrict__ num, const T* __restrict__ den,
                              T eps, T* __restrict__ out, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  out[idx] = s[idx] * (num[idx] / (den[idx] + eps));
}


// B = A^T on n x n.
template <typename T>
__global__ void kxk_transpose_kernel(const T* __restrict__ A, T* __restrict__ B, int n) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  B[idx] = A[(idx % n) * n + idx / n];
}

// The k x k chains of an NMTF iteration in one launch each (n <= 64): a single workgroup walks the stages, staging
// the two operands of a stage in LDS.  Every element is the same ascending-l sum as in kxk_gemm_kernel, so the results
// are bit-identical to the separate launches -- the gain is the launches themselves (6-9 us each for matrices that
// take a microsecond), a quarter of an iteration at 1e7 nnz.
template <typename T>
__device__ __forceinline__ void kxk_stage(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C, int n,
                                          int ta, int tb, T* la, T* lb) {
  __syncthreads();   // the previous stage's output is written; its LDS images are free
  for (int i = threadIdx.x; i < n * n; i += blockDim.x) {
    la[i] = A[i];
    lb[i] = B[i];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
    const int i = idx / n, jx = idx % n;
    T s = T(0);
    for (int l = 0; l < n; ++l) {
      const T av = ta ? la[l * n + i] : la[i * n + l];
      const T bv = tb ? lb[jx * n + l] : lb[l * n + jx];
      s += av * bv;
    }
    C[idx] = s;
  }
}

// before the W update (nmtf.py:151-152): ST = s^T, T1 = s (h hT), Dm = T1 s^T
template <typename T>
__global__ __launch_bounds__(1024) void nmtf_chain_w_kernel(const T* __restrict__ S, const T* __restrict__ hhT,
                                                            T* __restrict__ ST, T* __restrict__ T1, T* __restrict__ Dm, int n) {
  extern __shared__ __align__(16) unsigned char chain_smem[];
  T* la = reinterpret_cast<T*>(chain_smem);
  T* lb = la + n * n;
  for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) ST[idx] = S[(idx % n) * n + idx / n];
  kxk_stage(S, hhT, T1, n, 0, 0, la, lb);
  kxk_stage(T1, S, Dm, n, 0, 1, la, lb);
}

// S update and the H update's k x k factor (nmtf.py:157-158, 162): T2 = (w1T w1) s, den = T2 (h hT),
// s1 = s * (num / (den + eps)), T3 = s1^T (w1T w1), Q = T3 s1
template <typename T>
__global__ __launch_bounds__(1024) void nmtf_chain_s_kernel(const T* __restrict__ Gw1, const T* __restrict__ S,
                                                            const T* __restrict__ hhT, const T* __restrict__ num, T eps,
                                                            T* __restrict__ T2, T* __restrict__ den, T* __restrict__ Sn,
                                                            T* __restrict__ T3, T* __restrict__ Qm, int n) {
  extern __shared__ __align__(16) unsigned char chain_smem[];
  T* la = reinterpret_cast<T*>(chain_smem);
  T* lb = la + n * n;
  kxk_stage(Gw1, S, T2, n, 0, 0, la, lb);
  kxk_stage(T2, hhT, den, n, 0, 0, la, lb);
  __syncthreads();
  for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) Sn[idx] = S[idx] * (num[idx] / (den[idx] + eps));
  kxk_stage(Sn, Gw1, T3, n, 1, 0, la, lb);
  kxk_stage(T3, Sn, Qm, n, 0, 0, la, lb);
}

// NMTF topic-relationship summary (s_matrix_confusion, nmtf.py:62-80): for every column i of S (k x k, leading
// dimension lds) the row of its maximum (first one on ties, as np.argmax) and that value; the columns are then
// ranked by value, descending (equal values: larger column first = the reverse of a stable ascending argsort).
// One workgroup; thread i owns column i (k <= blockDim.x).  Outputs are in rank order.
template <typename T>
__global__ void s_confusion_kernel(const T* __restrict__ S, int k, int lds, int32_t* __restrict__ col_out,
                                   int32_t* __restrict__ row_out, T* __restrict__ val_out) {
  extern __shared__ unsigned char s_conf_smem[];
  T* best = reinterpret_cast<T*>(s_conf_smem);
  const int i = threadIdx.x;
  T v = T(0);
  int arg = 0;
  if (i < k) {
    v = S[i];
    for (int r = 1; r < k; ++r) {
      const T x = S[(int64_t)r * lds + i];
      // np.argmax (nmtf.py:69): the first NaN of a column wins, and nothing replaces it
      if (x > v || (x != x && v == v)) {
        v = x;
        arg = r;
      }
    }
    best[i] = v;
  }
  __syncthreads();
  if (i < k) {
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const T x = best[j];
      // np.argsort(...)[::-1] (nmtf.py:73): NaN ranks as the greatest value; equal values (two NaNs included) put the
      // larger column first -- a total order, so every rank is written exactly once
      const bool xn = x != x, vn = v != v;
      const bool greater = (xn && !vn) || (!xn && !vn && x > v);
      const bool equal = (xn && vn) || (!xn && !vn && x == v);
      rank += (greater || (equal && j > i)) ? 1 : 0;
    }
    col_out[rank] = i;
    row_out[rank] = arg;
    val_out[rank] = v;
  }
}

// k x k (dense) -> ld x ld zero padded.
template <typename T>
__global__ void pad_sq_kernel(const T* __restrict__ in, int k, int ld, T* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ld * ld) return;
  const int r = idx / ld, c = idx % ld;
  out[idx] = (r < k && c < k) ? in[r * k + c] : T(0);
}

// ------------------------------------------------------------------------------- layout conversion
// rows x k (dense, leading dim k) -> rows x ld zero padded.
// `map` (may be null): internal row id of user row r (the vocabulary renumbering).
template <typename T>
__global__ void pad_rows_kernel(const T* __restrict__ in, int64_t rows, int k, int ld, T* __restrict__ out,
                                const int32_t* __restrict__ map = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  const int64_t r = i / ld;
  const int c = (int)(i % ld);
  const int64_t ro = map ? map[r] : r;
  out[ro * ld + c] = c < k ? in[r * k + c] : T(0);
}

// rows x ld -> rows x k, entries below thr set to 0 (thr <= 0: plain copy) (nmf_basic.py:86-87).
// coff: first of the k columns inside the padded row (batched sweep; 0 otherwise).
template <typename T>
__global__ void unpad_rows_kernel(const T* __restrict__ in, int64_t rows, int k, int ld, T thr,
                                  T* __restrict__ out, const int32_t* __restrict__ map = nullptr, int coff = 0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * k) return;
  const int64_t r = i / k;
  const int c = (int)(i % k);
  const T v = in[(map ? (int64_t)map[r] : r) * ld + coff + c];
  out[i] = (thr > T(0) && v < thr) ? T(0) : v;
}

// Batched sweep, boundary only: columns [coff, coff + k) of the padded rows <- a rows x k matrix (transposed = 0) or
// the transpose of a k x rows matrix (transposed = 1: H as the reference holds it); the other columns are left alone.
template <typename T>
__global__ void put_cols_kernel(const T* __restrict__ in, int64_t rows, int k, int ld, int coff, int transposed,
                                T* __restrict__ out, const int32_t* __restrict__ map) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * k) return;
  const int64_t r = transposed ? i % rows : i / k;
  const int c = (int)(transposed ? i / rows : i % k);
  out[(map ? (int64_t)map[r] : r) * ld + coff + c] = in[i];
}

// ... and back: a k x rows matrix <- the transpose of columns [coff, coff + k), thresholded (H out).
template <typename T>
__global__ void get_cols_t_kernel(const T* __restrict__ in, int64_t rows, int k, int ld, int coff, T thr,
                                  T* __restrict__ out, const int32_t* __restrict__ map) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * k) return;
  const int64_t r = i % rows;
  const int c = (int)(i / rows);
  const T v = in[(map ? (int64_t)map[r] : r) * ld + coff + c];
  out[i] = (thr > T(0) && v < thr) ? T(0) : v;
}

// G (ld x ld) <- the k x k diagonal block of the concatenated Gram `cat` that starts at (coff, coff), zero padded.
template <typename T>
__global__ void diag_block_kernel(const T* __restrict__ cat, int ld, int coff, int k, T* __restrict__ G) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ld * ld) return;
  const int r = idx / ld, c = idx % ld;
  G[idx] = (r < k && c < k) ? cat[(coff + r) * ld + coff + c] : T(0);
}

// H (k x V row-major) -> Ht (V x ld, zero padded), 32 x 32 tiles through LDS.
template <typename T>
__global__ __launch_bounds__(256) void transpose_in_kernel(const T* __restrict__ H, int k, int64_t V, int ld,
                                                           T* __restrict__ Ht, const int32_t* __restrict__ map) {
  __shared__ T tile[32][33];
  const int64_t v0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const int64_t v = v0 + tx;
    tile[i][tx] = (c < k && v < V) ? H[(int64_t)c * V + v] : T(0);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int64_t v = v0 + i;
    const int c = c0 + tx;
    if (v < V && c < ld) Ht[(map ? (int64_t)map[v] : v) * ld + c] = tile[tx][i];
  }
}

// Ht (V x ld) -> H (k x V), thresholded.
template <typename T>
__global__ __launch_bounds__(256) void transpose_out_kernel(const T* __restrict__ Ht, int k, int64_t V, int ld,
                                                            T thr, T* __restrict__ H, const int32_t* __restrict__ map) {
  __shared__ T tile[32][33];
  const int64_t v0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int64_t v = v0 + i;
    const int c = c0 + tx;
    tile[i][tx] = (v < V && c < ld) ? Ht[(map ? (int64_t)map[v] : v) * ld + c] : T(0);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const int64_t v = v0 + tx;
    if (c < k && v < V) {
      const T x = tile[tx][i];
      H[(int64_t)c * V + v] = (thr > T(0) && x < thr) ? T(0) : x;
    }
  }
}

// in-place threshold of a padded matrix (pnmf: W[W < 1e-10] = 0, nmf_projective_basic.py:111).
template <typename T>
__global__ void threshold_kernel(T* __restrict__ a, int64_t n, T thr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (a[i] < thr) a[i] = T(0);
}

// inv[c] = 1 / norm_c with norm_c = sqrt(G[c][c]) and norms < 10*eps(T) replaced by 1
// (sklearn normalize(..., axis=0) as called at nmf_projective_basic.py:114).
template <typename T>
__global__ void inv_colnorm_kernel(const T* __restrict__ G, int ld, T tiny, T* __restrict__ inv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  T nrm = sqrt(G[(int64_t)c * ld + c]);
  if (nrm < tiny) nrm = T(1);
  inv[c] = nrm;  // the reference divides; keep the divisor, not its reciprocal
}

template <typename T>
__global__ void div_cols_kernel(T* __restrict__ a, int64_t rows, int ld, const T* __restrict__ nrm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  a[i] = a[i] / nrm[i % ld];
}

}  // namespace mnmf
