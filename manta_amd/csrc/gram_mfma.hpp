// Tall-skinny Gram products on the matrix cores (fp32):  G (ld x ld) = A^T . B over `rows` rows
// (`w.T @ w` nmf_basic.py:39, `h @ h.T` :44 on H^T, `W.T @ pay` nmf_projective_basic.py:75, and with DIFF
// the Gram of |new - old| behind the convergence norms, nmf_basic.py:52-53).
//
// v_mfma_f32_32x32x2_f32 computes D(32x32) += Aop(32x2) . Bop(2x32); with Aop[i][kk] = A[r+kk][32qi+i] and
// Bop[kk][j] = B[r+kk][32qj+j] one instruction adds two rows of the factor into one 32x32 tile of G.
// Lane l supplies A[r + (l>>5)][32qi + (l&31)]: the operand registers are filled straight from global
// memory with two fully coalesced 128-byte runs per wave load -- no LDS, no transposition.  The result is
// exact fp32 (a k-ordered fmaf chain).  Each of the 8 waves of a block owns TR x NT tiles and strides through
// the block's rows; the waves add their tiles into one LDS image in wave order (fixed order, no atomics) and
// the block writes one partial G to the slab, which slab_reduce_kernel sums in a fixed order.
#pragma once
#include "common.hpp"

namespace mnmf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NT, int TR, bool DIFF>
__global__ __launch_bounds__(512) void gram_mfma_f32_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                            const float* __restrict__ B, const float* __restrict__ B2,
                                                            int64_t rows, int ld, int64_t rpb,
                                                            float* __restrict__ slab) {
  constexpr int UR = 4;  // row pairs in flight
  constexpr int NW = 8;  // waves per block
  __shared__ float red[TR * NT * 16 * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cl = lane & 31, rh = lane >> 5;
  const int qi0 = blockIdx.y * TR;
  f32x16 acc[TR][NT];
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][q][e] = 0.f;
  const int64_t r_begin = (int64_t)blockIdx.x * rpb;
  const int64_t r_end = (r_begin + rpb < rows) ? r_begin + rpb : rows;
  bool aok[TR], bok[NT];
  int acol[TR], bcol[NT];
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    acol[t] = 32 * (qi0 + t) + cl;
    aok[t] = acol[t] < ld;
  }
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    bcol[q] = 32 * q + cl;
    bok[q] = bcol[q] < ld;
  }
  for (int64_t r0 = r_begin + 2 * wv; r0 < r_end; r0 += 2 * NW * UR) {
    float a[UR][TR], b[UR][NT];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t row = r0 + 2 * NW * u + rh;
      const bool rok = row < r_end;
      const int64_t off = row * ld;
#pragma unroll
      for (int t = 0; t < TR; ++t) {
        float x = 0.f;
        if (rok && aok[t]) {
          x = A[off + acol[t]];
          if (DIFF) x = fabsf(x - A2[off + acol[t]]);
        }
        a[u][t] = x;
      }
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        float x = 0.f;
        if (rok && bok[q]) {
          x = B[off + bcol[q]];
          if (DIFF) x = fabsf(x - B2[off + bcol[q]]);
        }
        b[u][q] = x;
      }
    }
#pragma unroll
    for (int u = 0; u < UR; ++u)
#pragma unroll
      for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q)
          acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][q], acc[t][q], 0, 0, 0);
  }
  // block reduction in wave order through LDS (element (t,q,e) of lane l at ((t*NT+q)*16+e)*64 + l)
  for (int w = 0; w < NW; ++w) {
    if (wv == w) {
#pragma unroll
      for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int idx = ((t * NT + q) * 16 + e) * 64 + lane;
            red[idx] = (w == 0) ? acc[t][q][e] : red[idx] + acc[t][q][e];
          }
    }
    __syncthreads();
  }
  // C/D layout of the 32x32 tile: lane holds column l&31, rows (e&3) + 8*(e>>2) + 4*(l>>5)
  float* dst = slab + (int64_t)blockIdx.x * (int64_t)ld * ld;
  for (int i = threadIdx.x; i < TR * NT * 16 * 64; i += NW * 64) {
    const int l = i & 63, e = (i >> 6) & 15, tq = i >> 10;
    const int t = tq / NT, q = tq % NT;
    const int gi = 32 * (qi0 + t) + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
    const int gj = 32 * q + (l & 31);
    if (gi < ld && gj < ld) dst[(int64_t)gi * ld + gj] = red[i];
  }
}

}  // namespace mnmf
