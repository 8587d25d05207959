// CSR SpMM on the stream layout:  out[r, :] = sum_p vals[p] * B[col[p], :]   (K1 and K2 of SURVEY 2a;
// `in_mat @ h.T` nmf_basic.py:45, `w.T @ in_mat` nmf_basic.py:40 computed as X^T . W).
//
// Lane mapping (wave64): a row of the dense factor is ld elements = ld/VEC 16-byte vectors.  A group
// of LPR lanes covers one factor row (lane li owns vector li, and vector LPR+li when VPL==2), so one
// wave-instruction gathers NG = 64/LPR factor rows, 16 B per lane, each row one contiguous read.
// Each group streams its own segment; the nnz (col,val) of a segment are fetched LPR at a time with
// one coalesced load per group (double-buffered a chunk ahead) and handed to the lanes by shuffles.
// A piece's sum is kept in registers and stored when its end flag passes (to the output row, or to a
// carry slot when the row consists of several pieces): no atomics, fixed order.
#pragma once
#include "common.hpp"

namespace mnmf {

// DIR only names the instantiation (0: X.B, 1: X^T.B) so profiles list the two directions apart.
template <typename T, int LPR, int VPL, int U, int DIR>
__global__ __launch_bounds__(256) void spmm_stream_kernel(
    const int64_t* __restrict__ seg_start, const int32_t* __restrict__ seg_len,
    const int32_t* __restrict__ seg_info, int64_t n_seg,
    const uint32_t* __restrict__ cols, const T* __restrict__ vals, const int32_t* __restrict__ dest,
    const T* __restrict__ B, int ld, T* __restrict__ out, T* __restrict__ carry) {
  constexpr int VEC = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  constexpr int NG = kWave / LPR;
  static_assert(LPR % U == 0, "U must divide LPR");
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const int gbase = gi * LPR;
  const int64_t seg = wave * NG + gi;
  const bool valid = seg < n_seg;
  int64_t p = 0, e = 0;
  int j = 0;  // index of the current piece in dest[]
  if (valid) {
    p = seg_start[seg];
    e = p + seg_len[seg];
    j = seg_info[seg];
  }
  int coff[VPL];
  bool colok[VPL];
#pragma unroll
  for (int w = 0; w < VPL; ++w) {
    coff[w] = (w * LPR + li) * VEC;
    colok[w] = coff[w] < ld;
  }
  T acc[VPL][VEC];
#pragma unroll
  for (int w = 0; w < VPL; ++w)
#pragma unroll
    for (int x = 0; x < VEC; ++x) acc[w][x] = T(0);

  uint32_t cl = 0;
  T vl = T(0);
  {
    const int64_t q = p + li;
    if (q < e) {
      cl = cols[q];
      vl = vals[q];
    }
  }
  while (__any(p < e)) {
    uint32_t cn = 0;
    T vn = T(0);
    {
      const int64_t q = p + LPR + li;
      if (q < e) {
        cn = cols[q];
        vn = vals[q];
      }
    }
#pragma unroll
    for (int u0 = 0; u0 < LPR; u0 += U) {
      uint32_t c[U];
      T v[U];
      VT b[U][VPL];
      int r[U];
      bool act[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        c[u] = (uint32_t)shfl_i((int)cl, gbase + u0 + u);
        v[u] = shfl_t(vl, gbase + u0 + u);
        act[u] = (p + u0 + u) < e;
      }
      int jj = j;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t col = (int64_t)(c[u] & ~kEndFlag);
        const T* bp = B + col * ld;
#pragma unroll
        for (int w = 0; w < VPL; ++w) {
          if (act[u] && colok[w]) {
            b[u][w] = *reinterpret_cast<const VT*>(bp + coff[w]);
          } else {
#pragma unroll
            for (int x = 0; x < VEC; ++x) b[u][w].v[x] = T(0);
          }
        }
        const bool endf = act[u] && (c[u] & kEndFlag);
        r[u] = 0;
        if (endf) r[u] = dest[jj];
        jj += endf ? 1 : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int w = 0; w < VPL; ++w)
#pragma unroll
          for (int x = 0; x < VEC; ++x) acc[w][x] += v[u] * b[u][w].v[x];
        if (act[u] && (c[u] & kEndFlag)) {
          // destination of the finished piece: an output row, or a carry slot of a multi-piece row
          T* op = r[u] >= 0 ? out + (int64_t)r[u] * ld : carry + (int64_t)(~r[u]) * ld;
#pragma unroll
          for (int w = 0; w < VPL; ++w) {
            if (colok[w]) {
              VT o;
#pragma unroll
              for (int x = 0; x < VEC; ++x) o.v[x] = acc[w][x];
              *reinterpret_cast<VT*>(op + coff[w]) = o;
            }
#pragma unroll
            for (int x = 0; x < VEC; ++x) acc[w][x] = T(0);
          }
        }
      }
      j = jj;
    }
    p = (p + LPR < e) ? p + LPR : e;
    cl = cn;
    vl = vn;
  }
}

// One wave per chunked row: out[row] = sum of its carry slots in a fixed order.
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const int32_t* __restrict__ fix_row,
                                                         const int32_t* __restrict__ fix_first,
                                                         const int32_t* __restrict__ fix_count, int64_t n_fix,
                                                         const T* __restrict__ carry, int ld,
                                                         T* __restrict__ out) {
  constexpr int VEC = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  constexpr int NG = kWave / LPR;
  const int lane = threadIdx.x & 63;
  const int64_t item = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (item >= n_fix) return;  // wave-uniform
  const int gi = lane / LPR, li = lane % LPR;
  const int64_t first = fix_first[item];
  const int n = fix_count[item];
  T acc[VPL][VEC];
#pragma unroll
  for (int w = 0; w < VPL; ++w)
#pragma unroll
    for (int x = 0; x < VEC; ++x) acc[w][x] = T(0);
  for (int s = gi; s < n; s += NG) {
    const T* cp = carry + (first + s) * ld;
#pragma unroll
    for (int w = 0; w < VPL; ++w) {
      const int co = (w * LPR + li) * VEC;
      if (co < ld) {
        const VT t = *reinterpret_cast<const VT*>(cp + co);
#pragma unroll
        for (int x = 0; x < VEC; ++x) acc[w][x] += t.v[x];
      }
    }
  }
#pragma unroll
  for (int mask = LPR; mask < kWave; mask <<= 1)
#pragma unroll
    for (int w = 0; w < VPL; ++w)
#pragma unroll
      for (int x = 0; x < VEC; ++x) acc[w][x] += shfl_xor_t(acc[w][x], mask);
  if (gi == 0) {
    T* op = out + (int64_t)fix_row[item] * ld;
#pragma unroll
    for (int w = 0; w < VPL; ++w) {
      const int co = (w * LPR + li) * VEC;
      if (co < ld) {
        VT o;
#pragma unroll
        for (int x = 0; x < VEC; ++x) o.v[x] = acc[w][x];
        *reinterpret_cast<VT*>(op + co) = o;
      }
    }
  }
}

template <typename T>
__global__ void zero_rows_kernel(const int32_t* __restrict__ rows, int64_t n, int ld, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ld) return;
  out[(int64_t)rows[i / ld] * ld + (i % ld)] = T(0);
}

}  // namespace mnmf
