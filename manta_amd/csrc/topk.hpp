// Per-topic rankings on the device: the n largest entries of every row (H: words of a topic,
// `np.flip(np.argsort(H[i, :]))[:n]`, topic_extractor.py:294-300) or of every column (W: documents of a topic,
// `W[:, i]`) without downloading the factor.
//
// Selection instead of sorting: pass j picks the largest element that is strictly smaller than pass j-1's pick in
// the total order (value, index) -- on equal values the larger index ranks first, which is what flipping a stable
// ascending argsort gives.  n is small (10-20), a pass is one coalesced scan (rows) or one strided scan (columns),
// the block reduces its candidates by shuffles and through LDS in a fixed order: deterministic.  NaN ranks last.
#pragma once
#include "common.hpp"

namespace mnmf {

template <typename T>
__device__ inline bool topk_less(T av, int64_t ai, T bv, int64_t bi) {  // (av, ai) < (bv, bi)
  return av < bv || (av == bv && ai < bi);
}

// grid = groups (rows, or columns when by_column); block 256.  A is rows x cols row-major.
template <typename T>
__global__ __launch_bounds__(256) void topk_kernel(const T* __restrict__ A, int64_t rows, int64_t cols, int by_column,
                                                   int n, int32_t* __restrict__ idx_out, T* __restrict__ val_out) {
  __shared__ T sv[4];
  __shared__ int64_t si[4];
  __shared__ T pick_v;
  __shared__ int64_t pick_i;
  const int64_t g = blockIdx.x;
  const int64_t len = by_column ? rows : cols;
  const int64_t stride = by_column ? cols : 1;
  const T* base = by_column ? A + g : A + g * cols;
  const T lowest = -INFINITY;
  T pv = INFINITY;          // previous pick: everything is below (+inf, len)
  int64_t pi = len;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int j = 0; j < n; ++j) {
    T bv = lowest;
    int64_t bi = -1;
    for (int64_t e = threadIdx.x; e < len; e += blockDim.x) {
      T v = base[e * stride];
      if (!(v == v)) v = lowest;                                   // NaN ranks last
      if (topk_less(v, e, pv, pi) && (bi < 0 || topk_less(bv, bi, v, e))) {
        bv = v;
        bi = e;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const T ov = shfl_t(bv, lane ^ off);
      const int olo = shfl_i((int)(bi & 0xffffffff), lane ^ off), ohi = shfl_i((int)(bi >> 32), lane ^ off);
      const int64_t oi = ((int64_t)ohi << 32) | (uint32_t)olo;
      if (oi >= 0 && (bi < 0 || topk_less(bv, bi, ov, oi))) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      sv[wv] = bv;
      si[wv] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      T fv = sv[0];
      int64_t fi = si[0];
      for (int w = 1; w < 4; ++w)
        if (si[w] >= 0 && (fi < 0 || topk_less(fv, fi, sv[w], si[w]))) {
          fv = sv[w];
          fi = si[w];
        }
      pick_v = fv;
      pick_i = fi;
      idx_out[g * n + j] = (int32_t)fi;                            // -1 when the group has fewer than n entries
      val_out[g * n + j] = fi >= 0 ? fv : lowest;
    }
    __syncthreads();
    pv = pick_v;
    pi = pick_i;
    if (pi < 0) {                                                  // exhausted: the remaining slots stay -1
      for (int jj = j + 1 + threadIdx.x; jj < n; jj += blockDim.x) {
        idx_out[g * n + jj] = -1;
        val_out[g * n + jj] = lowest;
      }
      return;
    }
  }
}

}  // namespace mnmf
