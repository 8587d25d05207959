// Stream layout of a sparse matrix and the one-time preprocessing that builds it.
//
// A "stream" is a sequence of nnz (value + gather index) in PROCESSING order, made of PIECES: a piece is
// a run of nnz that one lane group accumulates into one destination, and the gather index of its last
// nnz carries kEndFlag in the top bit.  The destination of a piece is an output row (the piece is the
// whole row) or a CARRY slot (the row is made of several pieces: rows longer than kSubPiece nnz are cut,
// and rows of X^T are also cut at document-block boundaries so that consecutive pieces of DIFFERENT
// vocabulary rows gather the same L2-sized block of W rows).  A fix-up kernel adds a row's carry slots in
// a fixed order: deterministic, no atomics.  Consecutive pieces are packed into SEGMENTS of about seg_nnz
// nnz; the SpMM kernel gives one segment to each lane group of a wavefront, so every group streams a
// contiguous, equally sized run of nnz whatever the row-length distribution is.
#pragma once
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace mnmf {

constexpr int64_t kSubPiece = 512;  // longest piece (nnz)

struct HostPieces {             // in processing order; piece p covers nnz [start[p], start[p+1])
  std::vector<int64_t> start;   // n_pieces + 1
  std::vector<int32_t> row;     // output row of the piece
  std::vector<int32_t> block;   // optional (X^T): document block of the piece, -1 = not blocked
};

struct HostSegments {
  std::vector<int64_t> seg_start;  // n_seg (nnz offset of the segment's first nnz)
  std::vector<int32_t> seg_len;    // n_seg (nnz in the segment; 0 = padding)
  std::vector<int32_t> seg_info;   // index into dest of the segment's first piece
  std::vector<int32_t> dest;       // per sub-piece: >= 0 output row, < 0: ~carry slot
  std::vector<int64_t> piece_end;  // per sub-piece: position of its last nnz (gets kEndFlag)
  std::vector<int32_t> empty_rows, fix_row, fix_first, fix_count;
  int64_t n_carry = 0;
  int64_t n_seg() const { return (int64_t)seg_info.size(); }
};

// Pieces of a plain CSR matrix: one per non-empty row.
inline void pieces_from_rowptr(const int64_t* rp, int64_t rows, HostPieces& p) {
  p.start.clear();
  p.row.clear();
  for (int64_t r = 0; r < rows; ++r)
    if (rp[r + 1] > rp[r]) {
      p.start.push_back(rp[r]);
      p.row.push_back((int32_t)r);
    }
  p.start.push_back(rp[rows]);
}

// pieces (+ number of output rows) -> sub-pieces, destinations, carry table, segments.
// xcd_group > 0 (X^T with document blocks): segments are emitted so that the workgroups that share an
// XCD (workgroup id mod 8 under round-robin dispatch; speed only, never correctness) walk the document
// blocks b = x, x+8, ... in order, xcd_group segments per workgroup; lists are padded with empty segments.
// Tile stream (pc.block = tile of the piece, ascending): every tile's nnz are cut into `groups` segments of equal
// length (one per lane group of the persistent workgroup, so the groups of a tile finish together); pieces are cut
// at segment boundaries (each part gets its own carry slot).  tile_first[t] = first segment of tile t.  Rows without
// pieces are NOT listed as empty: they belong to the other X^T stream.
inline void build_tile_segments_host(const HostPieces& pc, int64_t rows, int64_t n_tiles, int groups, int64_t sub,
                                     HostSegments& s, std::vector<int32_t>& tile_first) {
  s = HostSegments();
  if (groups < 1) groups = 1;
  if (sub < 1) sub = 1;
  const int64_t np = (int64_t)pc.row.size();
  tile_first.assign((size_t)n_tiles + 1, 0);
  // pass 1: cut points; pass 2 needs the number of parts per row for the carry table
  struct Part { int64_t a, b; int32_t row; };
  std::vector<Part> parts;
  std::vector<int64_t> seg_first_part;  // per segment: index of its first part
  int64_t p = 0;
  for (int64_t t = 0; t < n_tiles; ++t) {
    tile_first[t] = (int32_t)s.seg_start.size();
    const int64_t p0 = p;
    while (p < np && pc.block[p] == t) ++p;
    if (p == p0) continue;
    const int64_t a0 = pc.start[p0], b0 = pc.start[p];
    const int64_t len = cdiv(b0 - a0, groups);
    int64_t q = p0;
    for (int64_t sa = a0; sa < b0; sa += len) {
      const int64_t sb = std::min(sa + len, b0);
      s.seg_start.push_back(sa);
      s.seg_len.push_back((int32_t)(sb - sa));
      seg_first_part.push_back((int64_t)parts.size());
      int64_t x = sa;
      while (x < sb) {
        while (pc.start[q + 1] <= x) ++q;
        const int64_t e = std::min(std::min(pc.start[q + 1], sb), x + sub);
        parts.push_back({x, e, pc.row[q]});
        x = e;
      }
    }
  }
  tile_first[n_tiles] = (int32_t)s.seg_start.size();
  std::vector<int32_t> cnt((size_t)rows, 0), first((size_t)rows, 0), used((size_t)rows, 0);
  for (const Part& pt : parts) ++cnt[pt.row];
  for (int64_t r = 0; r < rows; ++r)
    if (cnt[r] > 1) {
      first[r] = (int32_t)s.n_carry;
      s.fix_row.push_back((int32_t)r);
      s.fix_first.push_back((int32_t)s.n_carry);
      s.fix_count.push_back(cnt[r]);
      s.n_carry += cnt[r];
    }
  s.dest.reserve(parts.size());
  s.piece_end.reserve(parts.size());
  for (const Part& pt : parts) {
    s.dest.push_back(cnt[pt.row] > 1 ? ~(first[pt.row] + used[pt.row]++) : pt.row);
    s.piece_end.push_back(pt.b - 1);
  }
  s.seg_info.resize(seg_first_part.size());
  for (size_t i = 0; i < seg_first_part.size(); ++i) s.seg_info[i] = (int32_t)seg_first_part[i];
}

// tile_first != nullptr (tile stream, the default mode): segments of about seg_nnz nnz that are closed at tile
// boundaries and stay in stream order; tile_first[t] = first segment of tile t; no empty-row list.
inline void build_segments_host(const HostPieces& pc, int64_t rows, int64_t seg_nnz, int64_t sub, HostSegments& s,
                                int xcd_group = 0, std::vector<int32_t>* tile_first = nullptr, int64_t n_tiles = 0,
                                bool exact = false) {
  // exact: a segment is closed after exactly seg_nnz nnz (the piece is cut there and both parts get carry slots), so
  // the lane groups of a wave, which work on consecutive segments, finish together; otherwise a segment ends with the
  // (sub-)piece that takes it to seg_nnz or beyond, and is up to `sub` nnz longer.
  s = HostSegments();
  if (seg_nnz < 1) seg_nnz = 1;
  if (sub < 1) sub = 1;
  const int64_t np = (int64_t)pc.row.size();
  const bool tiled = tile_first != nullptr && pc.block.size() == pc.row.size();
  const bool blocked = (xcd_group > 0 || tiled) && pc.block.size() == pc.row.size();
  // pass 1: parts (sub-pieces) and segments in processing order
  struct Part { int64_t b; int32_t row; };
  std::vector<Part> parts;
  std::vector<int32_t> cnt((size_t)rows, 0);
  std::vector<int32_t> seg_block;
  int64_t cur = -1;
  int32_t cur_block = 0;
  for (int64_t p = 0; p < np; ++p) {
    const int32_t r = pc.row[p];
    const int32_t blk = blocked ? pc.block[p] : 0;
    if (blocked && cur >= 0 && blk != cur_block) {  // close the open segment at a block boundary
      s.seg_len.push_back((int32_t)(pc.start[p] - cur));
      cur = -1;
    }
    for (int64_t a = pc.start[p]; a < pc.start[p + 1];) {
      if (cur < 0) {
        cur = a;
        cur_block = blk;
        s.seg_start.push_back(a);
        s.seg_info.push_back((int32_t)parts.size());
        seg_block.push_back(blk);
      }
      int64_t b = std::min(a + sub, pc.start[p + 1]);
      if (exact) b = std::min(b, cur + seg_nnz);
      parts.push_back({b, r});
      ++cnt[r];
      if (b - cur >= seg_nnz) {
        s.seg_len.push_back((int32_t)(b - cur));
        cur = -1;
      }
      a = b;
    }
  }
  if (cur >= 0) s.seg_len.push_back((int32_t)(pc.start[np] - cur));
  // pass 2: carry table and destinations
  std::vector<int32_t> first((size_t)rows, 0);
  for (int64_t r = 0; r < rows; ++r) {
    if (cnt[r] == 0 && !tiled) s.empty_rows.push_back((int32_t)r);
    if (cnt[r] > 1) {
      first[r] = (int32_t)s.n_carry;
      s.fix_row.push_back((int32_t)r);
      s.fix_first.push_back((int32_t)s.n_carry);
      s.fix_count.push_back(cnt[r]);
      s.n_carry += cnt[r];
    }
  }
  std::vector<int32_t> used((size_t)rows, 0);
  s.dest.reserve(parts.size());
  s.piece_end.reserve(parts.size());
  for (const Part& pt : parts) {
    s.dest.push_back(cnt[pt.row] > 1 ? ~(first[pt.row] + used[pt.row]++) : pt.row);
    s.piece_end.push_back(pt.b - 1);
  }
  if (tiled) {
    tile_first->assign((size_t)n_tiles + 1, 0);
    const int64_t ns = s.n_seg();
    int64_t i = 0;
    for (int64_t t = 0; t <= n_tiles; ++t) {
      while (i < ns && seg_block[i] < t) ++i;
      (*tile_first)[t] = (int32_t)i;
    }
    return;
  }
  if (!blocked) return;
  // ---- XCD-affine order
  constexpr int kXcd = 8;
  const int64_t ns = s.n_seg();
  std::vector<std::vector<int64_t>> lists(kXcd);
  std::vector<int64_t> loose;
  for (int64_t i = 0; i < ns; ++i) {
    if (seg_block[i] < 0) loose.push_back(i); else lists[seg_block[i] % kXcd].push_back(i);
  }
  // unblocked segments (short rows) top the lists up, shortest list first
  size_t li = 0;
  while (li < loose.size()) {
    int best = 0;
    for (int x = 1; x < kXcd; ++x)
      if (lists[x].size() < lists[best].size()) best = x;
    lists[best].push_back(loose[li++]);
  }
  size_t L = 0;
  for (int x = 0; x < kXcd; ++x) L = std::max(L, lists[x].size());
  L = (size_t)round_up((int64_t)L, xcd_group);
  std::vector<int64_t> st2;
  std::vector<int32_t> len2, info2;
  st2.reserve(L * kXcd);
  len2.reserve(L * kXcd);
  info2.reserve(L * kXcd);
  for (size_t i = 0; i < L; i += xcd_group)
    for (int x = 0; x < kXcd; ++x)
      for (int t = 0; t < xcd_group; ++t) {
        const size_t q = i + t;
        if (q < lists[x].size()) {
          const int64_t id = lists[x][q];
          st2.push_back(s.seg_start[id]);
          len2.push_back(s.seg_len[id]);
          info2.push_back(s.seg_info[id]);
        } else {
          st2.push_back(0);
          len2.push_back(0);
          info2.push_back(0);
        }
      }
  s.seg_start.swap(st2);
  s.seg_len.swap(len2);
  s.seg_info.swap(info2);
}

// ---------------------------------------------------------------- device preprocessing kernels
static __global__ void set_flags_at_kernel(const int64_t* __restrict__ pos, int64_t n, uint32_t* cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cols[pos[i]] |= kEndFlag;
}

static __global__ void clear_flags_kernel(uint32_t* cols, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cols[i] &= ~kEndFlag;
}

// rowid[p] = row that owns nnz p (binary search over the row pointer); iota[p] = p.
static __global__ void expand_rowids_kernel(const int64_t* __restrict__ rp, int64_t rows, int64_t nnz,
                                     uint32_t* __restrict__ rowid, uint32_t* __restrict__ iota) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  int64_t lo = 0, hi = rows;  // largest r with rp[r] <= p
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rp[mid] <= p) lo = mid; else hi = mid;
  }
  rowid[p] = (uint32_t)lo;
  iota[p] = (uint32_t)p;
}

// Sweep layout of the rare words (spmm.hpp, spmm_sweep_kernel): which words, where their accumulators live.
struct SweepMap {
  const int32_t* rare_idx;  // [V] dense index of a rare word among the rare words, -1 for the others; null: no sweep
  uint32_t P;               // partitions (= workgroups of the sweep kernel)
  uint32_t rpp;             // accumulator rows per partition
  uint32_t win_shift;       // log2(documents per window)
  uint32_t base;            // first key of the sweep stream
};
// dense index i of a rare word -> (partition, accumulator row): the rare words are dealt to the partitions in snake
// order (0 .. P-1, P-1 .. 0, ...): their ids are sorted by document frequency, so the partitions' nnz come out even
__host__ __device__ inline void sweep_place(uint32_t i, uint32_t P, uint32_t& part, uint32_t& row) {
  row = i / P;
  const uint32_t pos = i - row * P;
  part = (row & 1u) ? P - 1u - pos : pos;
}

// Sort key of nnz p for the transposed streams.  Words below n_hot (the most frequent ones after the engine's
// renumbering) go to the TILE stream: key = tile * n_hot + word with tile = document / tile_rows, all below hot_span.
// Rare words (sw.rare_idx >= 0) go to the SWEEP stream, behind everything else: key = sw.base + cell * rpp + row with
// cell = window * P + partition.  The others: hot_span + (document block, word); vocabulary rows with at most
// `long_df` nnz are not blocked (pseudo-block n_blocks), so they stay one piece.
static __global__ void transpose_key_kernel(const uint32_t* __restrict__ cols, const uint32_t* __restrict__ rowid,
                                     const int32_t* __restrict__ df, int64_t nnz, int64_t V, int block_shift,
                                     uint32_t n_blocks, int32_t long_df, uint32_t n_hot, uint32_t tile_rows,
                                     uint32_t hot_span, SweepMap sw, uint32_t* __restrict__ key) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  const uint32_t c = cols[p];
  if (c < n_hot) {
    key[p] = (rowid[p] / tile_rows) * n_hot + c;
    return;
  }
  if (sw.rare_idx != nullptr && sw.rare_idx[c] >= 0) {
    uint32_t part, row;
    sweep_place((uint32_t)sw.rare_idx[c], sw.P, part, row);
    key[p] = sw.base + ((rowid[p] >> sw.win_shift) * sw.P + part) * sw.rpp + row;
    return;
  }
  const uint32_t b = df[c] > long_df ? (rowid[p] >> block_shift) : n_blocks;
  key[p] = hot_span + (uint32_t)((uint64_t)b * (uint64_t)V + c);
}

// rare[c] = 1 for the words of the sweep stream: not in the tile stream, more than lo_df (>= 0) and at most long_df nnz
static __global__ void rare_mark_kernel(const int32_t* __restrict__ df, int64_t V, uint32_t n_hot, int32_t lo_df, int32_t long_df,
                                        int32_t* __restrict__ rare, int32_t* __restrict__ rare_df) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= V) return;
  const int r = (c >= n_hot && df[c] > lo_df && df[c] <= long_df) ? 1 : 0;
  rare[c] = r;
  rare_df[c] = r ? df[c] : 0;
}
// rare_idx: exclusive scan of rare[] -> -1 for the others; words[row * P + part] = c (pre-set to -1)
static __global__ void rare_place_kernel(const int32_t* __restrict__ rare, int32_t* __restrict__ rare_idx, int64_t V,
                                         uint32_t P, int32_t* __restrict__ words) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= V) return;
  if (!rare[c]) {
    rare_idx[c] = -1;
    return;
  }
  uint32_t part, row;
  sweep_place((uint32_t)rare_idx[c], P, part, row);
  words[(int64_t)row * P + part] = (int32_t)c;
}
// cell_start[c] = first position of the (sorted) sweep keys with key >= base + c * rpp, c = 0 .. n_cells
static __global__ void sweep_cells_kernel(const uint32_t* __restrict__ keys, int64_t n, uint32_t base, uint32_t rpp,
                                          int64_t n_cells, int32_t* __restrict__ cell_start) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_cells) return;
  const uint64_t want = (uint64_t)base + (uint64_t)c * rpp;
  int64_t lo = 0, hi = n;  // first index with keys[i] >= want
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((uint64_t)keys[mid] < want) lo = mid + 1; else hi = mid;
  }
  cell_start[c] = (int32_t)lo;
}
// seg[cell * ngw + g] = start of the g-th of ngw near-equal parts of the cell, moved forward to the next word boundary
// (a word of a window is accumulated by one lane group); seg[n_cells * ngw] = n.
static __global__ void sweep_segs_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ cell_start,
                                         int64_t n_cells, int ngw, int32_t* __restrict__ seg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_cells * ngw) return;
  if (i == n_cells * ngw) {
    seg[i] = cell_start[n_cells];
    return;
  }
  const int64_t c = i / ngw;
  const int g = (int)(i - c * ngw);
  const int32_t cs = cell_start[c], ce = cell_start[c + 1];
  int32_t t = cs + (int32_t)(((int64_t)(ce - cs) * g) / ngw);
  while (t > cs && t < ce && keys[t] == keys[t - 1]) ++t;
  seg[i] = t;
}

// flag[p] = 1 where a new piece starts (key differs from the previous nnz).
static __global__ void piece_flag_kernel(const uint32_t* __restrict__ key, int64_t nnz, uint8_t* __restrict__ flag) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  flag[p] = (p == 0 || key[p] != key[p - 1]) ? 1 : 0;
}

// order[i] = old id of the row with new id i  ->  newid[old] = i
static __global__ void invert_perm_kernel(const uint32_t* __restrict__ order, int64_t n, int32_t* __restrict__ newid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) newid[order[i]] = (int32_t)i;
}

// cols[p] = map[cols[p]]  (end flags must be clear)
static __global__ void remap_cols_kernel(uint32_t* cols, int64_t nnz, const int32_t* __restrict__ map) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < nnz) cols[p] = (uint32_t)map[cols[p]];
}

template <typename T>
__global__ void df_to_key_kernel(const int32_t* __restrict__ df, int64_t n, T* __restrict__ key, uint32_t* __restrict__ iota) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    key[i] = (T)df[i];
    iota[i] = (uint32_t)i;
  }
}

static __global__ void gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, int64_t n,
                                  uint32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

// sorted position i <- original nnz perm[i]; the first n_hot_nnz positions are the TILE stream (gather index local
// to the tile), the last n_sweep positions the SWEEP stream (document id + accumulator row, from the key), the rest
// the blocked stream
template <typename T>
__global__ void gather_transposed_kernel(const uint32_t* __restrict__ perm, const uint32_t* __restrict__ rowid,
                                         const T* __restrict__ vals, int64_t nnz, int64_t n_hot_nnz, uint32_t tile_rows,
                                         uint32_t* __restrict__ h_cols, T* __restrict__ h_vals,
                                         uint32_t* __restrict__ t_cols, T* __restrict__ t_vals,
                                         int64_t n_sweep = 0, const uint32_t* __restrict__ keys = nullptr, uint32_t sweep_base = 0,
                                         uint32_t rpp = 1, uint32_t* __restrict__ s_docs = nullptr,
                                         uint32_t* __restrict__ s_rows = nullptr, T* __restrict__ s_vals = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const uint32_t p = perm[i];
  if (i < n_hot_nnz) {
    h_cols[i] = rowid[p] % tile_rows;
    h_vals[i] = vals[p];
  } else if (i >= nnz - n_sweep) {
    const int64_t o = i - (nnz - n_sweep);
    s_docs[o] = rowid[p];
    s_rows[o] = (keys[i] - sweep_base) % rpp;
    s_vals[o] = vals[p];
  } else {
    t_cols[i - n_hot_nnz] = rowid[p];
    t_vals[i - n_hot_nnz] = vals[p];
  }
}

}  // namespace mnmf
