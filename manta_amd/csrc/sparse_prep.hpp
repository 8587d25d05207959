// Stream layout of a CSR matrix and the one-time preprocessing that builds it.
//
// A "stream" is the nnz sequence of a CSR matrix (values + column ids) where the column id of
// the LAST nnz of every ordinary row carries kEndFlag in its top bit, cut into SEGMENTS of about
// seg_nnz consecutive nnz that begin and end on row boundaries.  Rows longer than kLongRow carry
// no end flag; they are cut into chunk segments whose partial sums go to "carry" slots that a
// fix-up kernel adds in a fixed order (deterministic, no atomics).  The SpMM kernel gives one
// segment to each LPR-lane group of a wavefront, so every group streams a contiguous, equally
// sized piece of nnz whatever the row-length distribution is (documents: ~100 nnz; vocabulary
// rows of X^T: 1 .. m nnz).
#pragma once
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace mnmf {

struct HostSegments {
  std::vector<int64_t> seg_start;  // n_seg + 1
  std::vector<int32_t> seg_info;   // >=0: index into nzrow of first row; <0: ~carry slot
  std::vector<int32_t> nzrow, empty_rows, fix_row, fix_first, fix_count;
  int64_t n_carry = 0;
  int64_t n_seg() const { return (int64_t)seg_info.size(); }
};

// rowptr[rows+1] (rowptr[0]==0) -> segments.  seg_nnz: target nnz per segment; long_row: rows
// with more nnz than this are chunked into pieces of max(seg_nnz, 128).
inline void build_segments_host(const int64_t* rp, int64_t rows, int64_t seg_nnz, int64_t long_row,
                                HostSegments& s) {
  s = HostSegments();
  if (seg_nnz < 1) seg_nnz = 1;
  const int64_t chunk = std::max<int64_t>(seg_nnz, 128);
  int64_t cur = -1;
  int32_t cur_j = 0;
  for (int64_t r = 0; r < rows; ++r) {
    const int64_t a = rp[r], b = rp[r + 1], len = b - a;
    if (len == 0) {
      s.empty_rows.push_back((int32_t)r);
      continue;
    }
    const int32_t j = (int32_t)s.nzrow.size();
    s.nzrow.push_back((int32_t)r);
    if (len > long_row) {
      if (cur >= 0) {
        s.seg_start.push_back(cur);
        s.seg_info.push_back(cur_j);
        cur = -1;
      }
      const int64_t nch = cdiv(len, chunk);
      s.fix_row.push_back((int32_t)r);
      s.fix_first.push_back((int32_t)s.n_carry);
      s.fix_count.push_back((int32_t)nch);
      for (int64_t c = 0; c < nch; ++c) {
        s.seg_start.push_back(a + c * chunk);
        s.seg_info.push_back(~(int32_t)(s.n_carry++));
      }
    } else {
      if (cur < 0) {
        cur = a;
        cur_j = j;
      }
      if (b - cur >= seg_nnz) {
        s.seg_start.push_back(cur);
        s.seg_info.push_back(cur_j);
        cur = -1;
      }
    }
  }
  if (cur >= 0) {
    s.seg_start.push_back(cur);
    s.seg_info.push_back(cur_j);
  }
  s.seg_start.push_back(rp[rows]);
}

// ---------------------------------------------------------------- device preprocessing kernels
// rows with 1..long_row nnz: flag their last column id.
__global__ void set_end_flags_kernel(const int64_t* __restrict__ rp, int64_t rows, uint32_t* cols,
                                     int64_t long_row) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t a = rp[r], b = rp[r + 1];
  if (b > a && (b - a) <= long_row) cols[b - 1] |= kEndFlag;
}

// rowid[p] = row that owns nnz p (binary search over the row pointer).
__global__ void expand_rowids_kernel(const int64_t* __restrict__ rp, int64_t rows, int64_t nnz,
                                     uint32_t* __restrict__ rowid, uint32_t* __restrict__ iota) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  int64_t lo = 0, hi = rows;  // find largest r with rp[r] <= p
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rp[mid] <= p) lo = mid; else hi = mid;
  }
  rowid[p] = (uint32_t)lo;
  iota[p] = (uint32_t)p;
}

// ptr[j] = first index in sorted keys with key >= j, j = 0..n_keys_domain (inclusive).
__global__ void lower_bound_ptr_kernel(const uint32_t* __restrict__ keys, int64_t nnz, int64_t domain,
                                       int64_t* __restrict__ ptr) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > domain) return;
  int64_t lo = 0, hi = nnz;  // first index with keys[idx] >= j
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < j) lo = mid + 1; else hi = mid;
  }
  ptr[j] = lo;
}

template <typename T>
__global__ void gather_transposed_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rowid,
                                         const T* __restrict__ vals, int64_t nnz, uint32_t* __restrict__ t_cols,
                                         T* __restrict__ t_vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const uint32_t p = perm[i];
  t_cols[i] = rowid[p];
  t_vals[i] = vals[p];
}

}  // namespace mnmf
