// fp64 symmetric Grams on the matrix cores: G = D^T D with D = A (w.T@w, h@h.T: nmf_basic.py:39,44) or D = |A - A2|
// (the norms' Gram, :52-53), for factors of up to 128 padded columns.  The fp64 counterpart of gram_mfma_f32_kernel.
//
// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4).B(4x16); lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15].  With
// the four k's of a step being four consecutive rows of the factor, both operands of the tile pair (ta, tb) are the
// natural load: lane (j, q) holds element [row q][column 16t + j] -- 16 lanes read 128 contiguous bytes of a row, no
// transposition, no LDS.  The result has column l & 15 (of tile tb) on the lane and rows (l >> 4) + 4*reg (of tile ta)
// in its 4 registers.  A wave keeps the NT(NT+1)/2 upper-triangle tiles; it loads P steps (4P rows) at a time, one
// group ahead of the group it multiplies.  Waves take groups of their block's rows in turn, blocks reduce through LDS
// in wave order and write one partial per block to the slab (mirror images included); slab_reduce_kernel sums the
// partials in a fixed order.  ld = 16 * NT.
#pragma once
#include "common.hpp"

namespace mnmf {

template <int NT, bool DIFF>
__global__ __launch_bounds__(256, 1) void gram_mfma_f64_kernel(const double* __restrict__ A, const double* __restrict__ A2,
                                                               int64_t rows, int64_t rpb, double* __restrict__ slab) {
  using f64x4 = __attribute__((ext_vector_type(4))) double;
  constexpr int NTRI = NT * (NT + 1) / 2;
  constexpr int P = NT > 4 ? (DIFF || NT >= 8 ? 1 : 2) : 4;   // steps (of 4 rows) per group: what the register file holds twice next to the tiles
  constexpr int ld = 16 * NT;
  __shared__ double red[NTRI * 4 * 64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int64_t r_lo = (int64_t)blockIdx.x * rpb;
  const int64_t r_hi = (r_lo + rpb < rows) ? r_lo + rpb : rows;
  const int64_t n_groups = (r_hi - r_lo + 4 * P - 1) / (4 * P);
  f64x4 acc[NTRI];
#pragma unroll
  for (int x = 0; x < NTRI; ++x)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[x][e] = 0.0;
  // the next group's operands stay raw loads until the group is multiplied (|a - a2| computed there): nothing in the
  // prefetch waits for memory
  double nxt[P][NT], nx2[DIFF ? P : 1][DIFF ? NT : 1];
  auto fetch = [&](int64_t g) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int64_t r = r_lo + g * (4 * P) + 4 * p + q;
      const bool ok = r < r_hi;
      const double* a = A + r * ld + j;
      const double* a2 = DIFF ? A2 + r * ld + j : nullptr;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        nxt[p][t] = ok ? a[16 * t] : 0.0;
        if (DIFF) nx2[p][t] = ok ? a2[16 * t] : 0.0;
      }
    }
  };
  int64_t g = wv;
  if (g < n_groups) fetch(g);
#pragma unroll 1
  for (; g < n_groups; g += 4) {
    double cur[P][NT];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int t = 0; t < NT; ++t) cur[p][t] = DIFF ? fabs(nxt[p][t] - nx2[p][t]) : nxt[p][t];
    if (g + 4 < n_groups) fetch(g + 4);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      int y = 0;
#pragma unroll
      for (int ta = 0; ta < NT; ++ta)
#pragma unroll
        for (int tb = ta; tb < NT; ++tb, ++y)
          acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[p][ta], cur[p][tb], acc[y], 0, 0, 0);
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wv == w) {
#pragma unroll
      for (int x = 0; x < NTRI; ++x)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int idx = (x * 4 + e) * 64 + lane;
          red[idx] = (w == 0) ? acc[x][e] : red[idx] + acc[x][e];
        }
    }
    __syncthreads();
  }
  double* dst = slab + (int64_t)blockIdx.x * (int64_t)ld * ld;
  for (int i = threadIdx.x; i < NTRI * 4 * 64; i += 256) {
    const int l = i & 63, e = (i >> 6) & 3, x = i >> 8;
    int ta = 0, tb = x;  // x -> (ta, tb), ta <= tb, row-major over the upper triangle
    while (tb >= NT - ta) { tb -= NT - ta; ++ta; }
    tb += ta;
    const int gi = 16 * ta + (l >> 4) + 4 * e;
    const int gj = 16 * tb + (l & 15);
    const double v = red[i];
    dst[(int64_t)gi * ld + gj] = v;
    if (tb > ta) dst[(int64_t)gj * ld + gi] = v;
  }
}

}  // namespace mnmf
