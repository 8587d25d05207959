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

#ifndef MNMF_ABLATE
#define MNMF_ABLATE 0
#endif
// Non-temporal loads of the nnz stream were measured slower (C2: X^T.W 2.03 -> 2.16 ms): every 64-byte
// group read is half a line and the second half no longer hits; kept only as an experiment switch.
#ifndef MNMF_STREAM_NT
#define MNMF_STREAM_NT 0
#endif
#if MNMF_STREAM_NT
#define MNMF_NT_LOAD(p) __builtin_nontemporal_load(p)
#else
#define MNMF_NT_LOAD(p) (*(p))
#endif

#ifndef MNMF_COLD_NT
#define MNMF_COLD_NT 0   // experiment switch: 1 = every cold gather of the HOT mode non-temporal, 2 = only the tail rows
#endif
#ifndef MNMF_NT_STORE
#define MNMF_NT_STORE 0
#endif
#ifndef MNMF_TILE_CH
#define MNMF_TILE_CH 32   // nnz per stream chunk in the TILE mode (prefetch distance)
#endif
#ifndef MNMF_HOT_EPL
#define MNMF_HOT_EPL 1    // chunk length of the HOT / PLAIN modes in units of max(16, LPR) nnz
#endif
#ifndef MNMF_HOT_U
#define MNMF_HOT_U 8      // gathers per batch in the HOT mode (one vector per lane)
#endif
#ifndef MNMF_PLAIN_EPL
#define MNMF_PLAIN_EPL 1
#endif

namespace mnmf {

// The kernels fetch (index, value) and destination lists a whole chunk at a time, without bounds tests: every stream
// array (cols, vals, dest) is allocated with this many elements of slack behind its last entry.
constexpr size_t kStreamPad = 512;

struct alignas(16) SegDesc {
  int64_t start;   // first nnz of the segment
  int32_t len;     // nnz in the segment (0 = padding)
  int32_t info;    // index into dest of the segment's first piece
};

template <typename T>
struct SpmmArgs {
  const SegDesc* seg;        // [n_seg]
  int64_t n_seg;
  const uint32_t* cols;      // gather index, top bit = end of piece
  const T* vals;
  const int32_t* dest;       // per piece: >= 0 output row, < 0 ~carry slot
  int64_t n_dest;
  const T* B;                // gather table, tab_rows x ld
  int ld;                    // row stride of B, out and carry (elements)
  int nc;                    // columns handled by this launch (<= ld; < ld when a very wide factor is done in column chunks)
  T* out;
  T* carry;
  int hot_rows;              // HOT: first rows of B staged in LDS
  int lds_ld;                // HOT / TILE: LDS row stride (elements)
  uint32_t tab_bytes;        // bytes of B (non-WIDE: < 4 GiB)
  // TILE: the stream is cut into tiles of tile_rows consecutive rows of B; gather indices are local to the tile
  const int32_t* tile_first; // [n_tiles + 1] first segment of each tile
  int n_tiles;
  int tile_rows;
  int64_t tab_rows;          // rows of B
  int kreal;                 // columns that hold data (<= nc; the rest of a padded row of B is zero).  The LDS images
                             // (HOT, TILE) may be compact -- lds_ld = kreal rounded up to a vector, less than ld -- so
                             // that more rows fit; the lanes beyond then read into the next row and store zeros
};

// SPMM_QUEUE: the PLAIN kernel as a persistent launch (a few 256-thread workgroups per CU): like HOT without the LDS
// image, every lane group walks its own queue of segments back to back.  The four groups of a wave no longer wait for
// the longest of their four segments (X^T blocked stream of the mid-frequency words: pieces of up to 512 nnz make
// segments of 128..600 nnz, and 43 % of the wave-level gathers were issued for groups that had finished,
// TA_BUFFER_READ_WAVEFRONTS against nnz / 4), and the number of segments in flight is what fits the CUs.
enum SpmmMode { SPMM_PLAIN = 0, SPMM_HOT = 1, SPMM_TILE = 2, SPMM_QUEUE = 3 };

// The inner loop is written for a low instruction count per nnz (the gather of 16 B per lane is the
// only memory instruction per step): 32-bit offsets, one v_mad_u32_u24 per gather address (it ignores
// the flag bit), bounds-checked buffer loads, (col,val,dest) prefetched a chunk ahead with coalesced
// loads and handed out by ds_bpermute, and a wave-uniform branch around the rare end-of-piece work.
//
// DIR only names the instantiation (0: X.B, 1: X^T.B) so profiles list the two directions apart.
// WIDE: tables with >= 2^24 rows or >= 4 GiB use 64-bit addressing and plain global loads.
// HOT: persistent launch (one 1024-thread workgroup per CU, grid-stride over wave slots) that first
// stages the first `hot_rows` rows of B -- the most frequent vocabulary rows after the engine's
// document-frequency renumbering -- in LDS (row stride lds_ld, 16 B of padding against bank conflicts).
// Hot rows are read from LDS; only the lanes of cold rows issue the buffer load (an out-of-range offset
// instead of the exec mask was measured to cost the full vector-memory time).
// TILE (X^T.W for the most frequent words): persistent like HOT, but the workgroup walks tiles of tile_rows
// consecutive rows of B (documents): it stages the tile in LDS, then its lane groups process the tile's segments,
// whose gather indices are local to the tile -- every gather is an LDS read, the vector-memory path only carries
// the nnz stream and the carry stores.  Profiling showed both SpMMs bound by the texture-addresser (TA busy
// 80-93 %), which this takes out of the picture for about half of the nnz.
// Copy `rows` consecutive rows of a factor (row stride ld elements, contiguous in memory) into an LDS image with row
// stride lds_ld: padded (lds_ld >= ld, the whole row is copied) or compact (lds_ld < ld: only the first lds_ld
// elements of a row, the ones that hold data).  SU 16-byte buffer loads are in flight per thread before the first
// LDS write (one load at a time, each waited for, left a third of a tile's time to memory latency), on a descriptor of
// exactly those rows: past their end a load returns zeros, which are not stored.
template <typename T, int SU>
__device__ __forceinline__ void stage_rows(const T* __restrict__ src, int rows, int ld, int lds_ld, T* __restrict__ lds) {
  using U4 = __attribute__((ext_vector_type(4))) unsigned int;
  constexpr int VEC = VecOf<T>::N;
  const int nvr = ld / VEC;                               // vectors per row in memory
  const int nvc = (lds_ld < ld ? lds_ld : ld) / VEC;      // ... of them copied
  const int nvl = lds_ld / VEC;                           // ... per row of the image
  const int total = rows * nvc;
  const bool same = lds_ld == ld;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src), 0, (uint32_t)(rows * nvr) * 16u, 0x00020000);
  for (int i0 = threadIdx.x; i0 < total; i0 += blockDim.x * SU) {
    U4 t[SU];
    uint32_t o[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      uint32_t g = (uint32_t)i;
      o[u] = (uint32_t)i * 16u;
      if (!same) {
        const int r = i / nvc, c = i - r * nvc;
        g = (uint32_t)(r * nvr + c);
        o[u] = (uint32_t)(r * nvl + c) * 16u;
      }
      t[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i < total ? g * 16u : 0xfffffff0u, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u)
      if (i0 + u * (int)blockDim.x < total) *reinterpret_cast<U4*>(reinterpret_cast<char*>(lds) + o[u]) = t[u];
  }
}

template <typename T, int LPR, int VPL, int DIR, int MODE, bool WIDE>
#ifndef MNMF_HOT_BLOCK
#define MNMF_HOT_BLOCK 1024
#define MNMF_HOT_WAVES 4
#endif
__global__ __launch_bounds__(MODE == SPMM_HOT ? MNMF_HOT_BLOCK : MODE == SPMM_TILE ? 1024 : 256, MODE == SPMM_HOT ? MNMF_HOT_WAVES : 1) void spmm_stream_kernel(const SpmmArgs<T> a) {
  constexpr bool HOT = MODE == SPMM_HOT;
  constexpr bool TILE = MODE == SPMM_TILE;
  constexpr int VEC = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  using U4 = __attribute__((ext_vector_type(4))) unsigned int;
  constexpr int NG = kWave / LPR;
  // steps per chunk.  The next chunk of (index, value) is fetched while this one is processed; a step of the TILE
  // mode is an LDS read (tens of cycles), so its chunks are long enough to cover a trip to memory.
  constexpr int CH = (MODE == SPMM_TILE) ? LPR * (MNMF_TILE_CH / LPR > 8 ? 8 : (MNMF_TILE_CH / LPR < 1 ? 1 : MNMF_TILE_CH / LPR))
                                         : (LPR < 16 ? 16 : LPR) * (MODE == SPMM_HOT ? MNMF_HOT_EPL : MNMF_PLAIN_EPL);
  constexpr int EPL = CH / LPR;            // chunk elements held per lane
  constexpr int U = VPL == 2 ? 4 : (MODE == SPMM_HOT ? MNMF_HOT_U : 8);   // gathers in flight per batch
  static_assert(CH % U == 0, "U must divide CH");
  extern __shared__ __align__(16) unsigned char mnmf_smem[];
  T* lds = reinterpret_cast<T*>(mnmf_smem);
  const int ld = a.ld;
  if (HOT) {
    // the hot rows are the first rows of B, contiguous in memory: SU loads in flight per thread before the first LDS
    // write, on a descriptor of exactly those rows (past their end a load returns zeros, which are not stored)
    stage_rows<T, 6>(a.B, a.hot_rows, ld, a.lds_ld, lds);
    __syncthreads();
  }
#if MNMF_ABLATE == 5 || MNMF_ABLATE == 7   // the unstaged tile must at least hold finite numbers
  if (TILE) {
    for (int i = threadIdx.x; i < a.tile_rows * a.lds_ld; i += blockDim.x) lds[i] = T(1);
    __syncthreads();
  }
#endif
  const int lane = threadIdx.x & 63;
  const int gi = lane / LPR, li = lane % LPR;
  const int gbase = gi * LPR;
  bool colok[VPL], colreal[VPL];
  uint32_t cbytes[VPL];  // byte offset of this lane's vector in a row (clamped to 0 for lanes beyond the columns)
  uint32_t lbytes[VPL];  // the same for LDS reads, NOT clamped: idle lanes that all read offset 0 of their row put a
                         // second address on banks other lanes of the same ds_read_b128 lane group use (6 LDS cycles
                         // per instruction instead of 4 at ld = 112); reading on into the next row is conflict-free
                         // and harmless (the host sizes the LDS image with one vector row of slack)
#pragma unroll
  for (int w = 0; w < VPL; ++w) {
    const int coff = (w * LPR + li) * VEC;
    colok[w] = coff < a.nc;
    colreal[w] = coff < a.kreal;
    cbytes[w] = colok[w] ? (uint32_t)coff * (uint32_t)sizeof(T) : 0u;
    lbytes[w] = (uint32_t)coff * (uint32_t)sizeof(T);
  }
  const uint32_t rowbytes = (uint32_t)ld * (uint32_t)sizeof(T);
  const uint32_t lds_rowbytes = (uint32_t)a.lds_ld * (uint32_t)sizeof(T);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.B), 0, WIDE ? 0u : a.tab_bytes, 0x00020000);
  const int64_t n_outer = TILE ? a.n_tiles : 1;
  for (int64_t tile = TILE ? blockIdx.x : 0; tile < n_outer; tile += TILE ? gridDim.x : 1) {
  int64_t seg_lo = 0, seg_hi = a.n_seg;
  if (TILE) {
    __syncthreads();  // every wave is done with the previous tile
    const int64_t r0 = tile * a.tile_rows;
    const int nrow = (int)((a.tab_rows - r0 < a.tile_rows) ? a.tab_rows - r0 : a.tile_rows);
    const T* src = a.B + r0 * ld;
#if MNMF_ABLATE != 5 && MNMF_ABLATE != 7   // 5, 7: timing experiment only, the tile is not staged
    {
      // the tile is contiguous in memory (nrow rows of ld elements); SU loads are in flight per thread before the first
      // LDS write (one load at a time, each waited for, left a third of the tile's time to memory latency).  Buffer
      // loads on a descriptor of the tile itself: past its end they return zeros, which are not stored.
      stage_rows<T, 6>(src, nrow, ld, a.lds_ld, lds);
    }
#endif
    __syncthreads();
    seg_lo = a.tile_first[tile];
    seg_hi = a.tile_first[tile + 1];
  }
  // Segment queue of this lane group.  PLAIN: one segment per group and launch (the grid covers the segments).
  // HOT / TILE (persistent): group G of the NGT resident ones takes segments G, G + NGT, G + 2 NGT, ... of the tile
  // (TILE) or of the whole stream (HOT) and walks them BACK TO BACK: the chunk prefetch of a segment's last chunk
  // already fetches the first chunk of the group's next segment, whose descriptor was loaded two segments ahead.
  // Before, a wave started its NG next segments together: descriptor load, then first chunk, two exposed trips to
  // memory per ~100-500 nnz with 2-4 waves per SIMD to hide them (the waves of the tile kernel waited 58 % of their
  // cycles, profiles/r02_bench_c2_mfma_lds_summary.json), plus the idle tail of every group but the wave's longest.
  const int64_t gslot = TILE ? (int64_t)(threadIdx.x >> 6) : ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t gstride = MODE == SPMM_PLAIN ? 0 : (TILE ? (int64_t)(blockDim.x >> 6) : ((int64_t)gridDim.x * blockDim.x) >> 6) * NG;
  {
    int64_t seg = seg_lo + gslot * NG + gi;
    // descriptor of a segment (zero length past the end)
    int64_t s0 = 0, s0n = 0, s0nn = 0;
    int rem = 0, remn = 0, remnn = 0;   // nnz left in the current segment / length of the next two
    int j = 0, jn = 0, jnn = 0;         // index of the current piece in dest[] / first piece of the next two
    if (seg < seg_hi) {
      const SegDesc d = a.seg[seg];
      s0 = d.start, rem = d.len, j = d.info;
    }
    if (MODE != SPMM_PLAIN) {
      if (seg + gstride < seg_hi) {
        const SegDesc d = a.seg[seg + gstride];
        s0n = d.start, remn = d.len, jn = d.info;
      }
      if (seg + 2 * gstride < seg_hi) {
        const SegDesc d = a.seg[seg + 2 * gstride];
        s0nn = d.start, remnn = d.len, jnn = d.info;
      }
      seg += 2 * gstride;   // the segment whose descriptor was fetched last
    }
    const uint32_t* cp = a.cols + s0;
    const T* vp = a.vals + s0;
    T acc[VPL][VEC];
#pragma unroll
    for (int w = 0; w < VPL; ++w)
#pragma unroll
      for (int x = 0; x < VEC; ++x) acc[w][x] = T(0);
    // chunk registers: lane li holds elements li*EPL + x of the chunk (col 0 / val 0 beyond the segment:
    // such steps gather row 0, which is valid memory, and add 0).  The loads have no bounds test (kStreamPad), so the
    // EPL elements of a lane are one wide load; what lies beyond the segment is masked when the registers are taken over.
    uint32_t cl[EPL];
    T vl[EPL];
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
      const int q = li * EPL + x;
      cl[x] = MNMF_NT_LOAD(cp + q);
      vl[x] = MNMF_NT_LOAD(vp + q);
    }
    // destinations of the next CH pieces of this group: lane li holds dest[j + li*EPL + x].  Like (index, value)
    // they are fetched one chunk ahead: loaded at the start of the chunk that uses them, the first piece end of
    // every chunk waited for them with vmcnt(0) -- a full trip to memory, and in the PLAIN / HOT modes a drain of
    // every gather in flight.  Where the next chunk's list starts is known up front: j advances by the number of
    // end flags in the current chunk.
    int dl[EPL];
#pragma unroll
    for (int x = 0; x < EPL; ++x) dl[x] = a.dest[(int64_t)j + li * EPL + x];
    // Retire the first chunk's loads before the loop: otherwise hipcc, merging "load still pending" from
    // the loop entry with the back edge, puts a vmcnt(0) inside the loop right behind the prefetch of the
    // next chunk and every chunk stalls for a full memory latency.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
      const bool in = li * EPL + x < rem;
      cl[x] = in ? cl[x] : 0u;
      vl[x] = in ? vl[x] : T(0);
    }
    while (__any(rem > 0)) {
      // what this group works on after the current chunk: the rest of its segment, or its next segment
      const bool last = MODE != SPMM_PLAIN && rem <= CH;
      // descriptor of the segment after the next two, fetched with every chunk (one 16-byte load per lane group, the
      // first load of the iteration: behind a branch at the segment switch it would be waited for on the spot)
      SegDesc dx = SegDesc{0, 0, 0};
      if (MODE != SPMM_PLAIN) dx = a.seg[seg + gstride < seg_hi ? seg + gstride : seg_hi - 1];
      const uint32_t* cq = last ? a.cols + s0n : cp + CH;
      const T* vq = last ? a.vals + s0n : vp + CH;
      const int remq = last ? remn : rem - CH;
      uint32_t cn[EPL];
      T vn[EPL];
#pragma unroll
      for (int x = 0; x < EPL; ++x) {
        const int q = li * EPL + x;
        cn[x] = MNMF_NT_LOAD(cq + q);
        vn[x] = MNMF_NT_LOAD(vq + q);
      }
      const int jbase = j;
      // end-of-piece flags of the whole chunk as wave-level bit masks (bit = holder lane): the per-step
      // "did any group finish a piece" test is then scalar work
      uint64_t em[EPL];
#pragma unroll
      for (int x = 0; x < EPL; ++x) em[x] = __ballot((int)cl[x] < 0);
      uint64_t gmask = 0;  // bit 0 of every lane group
#pragma unroll
      for (int g = 0; g < NG; ++g) gmask |= (uint64_t)1 << (g * LPR);
      int dn[EPL];
      {
        int ends = 0;  // pieces of this group that end inside the current chunk
#pragma unroll
        for (int x = 0; x < EPL; ++x) {
          const uint64_t mine = LPR == 64 ? em[x] : (em[x] >> gbase) & (((uint64_t)1 << (LPR & 63)) - 1);
          ends += __popcll(mine);
        }
        const int64_t jq = last ? (int64_t)jn : (int64_t)j + ends;
#pragma unroll
        for (int x = 0; x < EPL; ++x) dn[x] = a.dest[jq + li * EPL + x];
      }
      static_for<CH / U>([&](auto ib) {
        constexpr int u0 = decltype(ib)::value * U;
        uint32_t c[U];
        T v[U];
        VT b[U][VPL];
        // hand out (col, val) of step u0+u: lane (u0+u)/EPL of the group holds it in register (u0+u)%EPL
        static_for<U>([&](auto iu) {
          constexpr int u = decltype(iu)::value;
          constexpr int t = u0 + u;
          c[u] = (uint32_t)group_bcast_i<LPR, t / EPL>((int)cl[t % EPL], lane);
          v[u] = group_bcast_t<LPR, t / EPL>(vl[t % EPL], lane);
        });
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (WIDE) {
            const T* bp = a.B + (int64_t)(c[u] & ~kEndFlag) * ld;
#pragma unroll
            for (int w = 0; w < VPL; ++w) b[u][w] = *reinterpret_cast<const VT*>(reinterpret_cast<const char*>(bp) + cbytes[w]);
          } else if (TILE) {
            const uint32_t lrow = __umul24(c[u], lds_rowbytes);  // u24: the flag bit drops out
#pragma unroll
            for (int w = 0; w < VPL; ++w) {
#if MNMF_ABLATE == 2 || MNMF_ABLATE == 4   // timing experiment only: no LDS read
              const uint32_t la = lrow + lbytes[w];
              asm volatile("" ::"v"(la));
              const U4 l = {la, 1u, 2u, 3u};
#else
              const U4 l = *reinterpret_cast<const U4*>(reinterpret_cast<const char*>(lds) + lrow + lbytes[w]);
#endif
              __builtin_memcpy(&b[u][w], &l, sizeof(VT));
            }
          } else if (HOT) {
            // hot rows: LDS read (always issued, cold lanes read row 0); cold rows: buffer load under the
            // exec mask of the cold lanes only, into registers of its own (no load waits on another)
            const uint32_t col = c[u] & ~kEndFlag;
            const bool hot = col < (uint32_t)a.hot_rows;
            const uint32_t lrow = __umul24(hot ? col : 0u, lds_rowbytes);
#pragma unroll
            for (int w = 0; w < VPL; ++w) {
              const U4 l = *reinterpret_cast<const U4*>(reinterpret_cast<const char*>(lds) + lrow + lbytes[w]);
#if MNMF_COLD_NT == 2   // experiment: rows far down the frequency order are gathered with the non-temporal policy
              const bool tail = col >= (uint32_t)a.hot_rows * 16u;
              U4 g, g2;
              if (!hot && !tail) g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, __umul24(col, rowbytes) + cbytes[w], 0, 0);
              if (tail) g2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, __umul24(col, rowbytes) + cbytes[w], 0, 2);
              const U4 sel = hot ? l : (tail ? g2 : g);
#else
              U4 g;
              if (!hot) g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, __umul24(col, rowbytes) + cbytes[w], 0, MNMF_COLD_NT ? 2 : 0);
              const U4 sel = hot ? l : g;
#endif
              __builtin_memcpy(&b[u][w], &sel, sizeof(VT));
            }
          } else {
#pragma unroll
            for (int w = 0; w < VPL; ++w) {
              const uint32_t voff = __umul24(c[u], rowbytes) + cbytes[w];  // u24: the flag bit drops out
#if MNMF_ABLATE == 1   // timing experiment only: no gather
              asm volatile("" ::"v"(voff));
              U4 g = {voff, 1u, 2u, 3u};
#else
              const U4 g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
#endif
              __builtin_memcpy(&b[u][w], &g, sizeof(VT));
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int w = 0; w < VPL; ++w)
#pragma unroll
#if MNMF_ABLATE == 3 || MNMF_ABLATE == 4   // timing experiment only: the gathered vector is waited for, not multiplied
            for (int x = 0; x < VEC; ++x) asm volatile("" ::"v"(b[u][w].v[x]), "v"(v[u]));
#else
            for (int x = 0; x < VEC; ++x) acc[w][x] += v[u] * b[u][w].v[x];
#endif
          const int t = u0 + u;  // step in the chunk: held by lane t / EPL of each group, register t % EPL
          const uint64_t eb = em[t % EPL] >> (t / EPL);
          if (eb & gmask) {  // wave-uniform, scalar: some group finished a piece at this step
            const bool endf = (eb >> gbase) & 1;
            // Destination of each finished piece: entry (j - jbase) of its group's list, i.e. register idx / LPR of
            // lane idx % LPR of the group.  Picked with v_readlane under a scalar loop over the groups that
            // finished (usually one): a ds_bpermute here is an LDS operation, and waiting for it drains every
            // gather the TILE / HOT modes have in flight in the LDS pipe.
            int d = 0;
            uint64_t pend = eb & gmask;
            if (!TILE) {  // vector-memory gathers: nothing of theirs waits behind a ds_bpermute
              const int idx = j - jbase;   // entry idx of the group's list: register idx % EPL of lane idx / EPL
              d = shfl_i(dl[0], gbase + (idx / EPL));
#pragma unroll
              for (int x = 1; x < EPL; ++x) {
                const int dx = shfl_i(dl[x], gbase + (idx / EPL));
                d = (idx % EPL == x) ? dx : d;
              }
              pend = 0;
            }
            while (pend) {
              const int b = __builtin_ctzll(pend);  // first lane of the group
              pend &= pend - 1;
              const int idx = __builtin_amdgcn_readlane(j - jbase, b);  // < CH, uniform inside the group
              int dg = __builtin_amdgcn_readlane(dl[0], b + (idx / EPL));
#pragma unroll
              for (int x = 1; x < EPL; ++x) {
                const int dx = __builtin_amdgcn_readlane(dl[x], b + (idx / EPL));
                dg = (idx % EPL == x) ? dx : dg;
              }
              d = gbase == b ? dg : d;
            }
            if (endf) {
              // destination of the finished piece: an output row, or a carry slot of a multi-piece row
              T* op = d >= 0 ? a.out + (int64_t)d * ld : a.carry + (int64_t)(~d) * ld;
#pragma unroll
              for (int w = 0; w < VPL; ++w) {
#if MNMF_ABLATE == 6 || MNMF_ABLATE == 7   // timing experiment only: finished pieces are not stored
                if (colok[w] && d == 0x7fffffff) {
#else
                if (colok[w]) {
#endif
                  VT o;
#pragma unroll
                  for (int x = 0; x < VEC; ++x) o.v[x] = colreal[w] ? acc[w][x] : T(0);
#if MNMF_NT_STORE   // experiment: output rows and carries are written once and read by another kernel
                  U4 ou;
                  __builtin_memcpy(&ou, &o, sizeof(VT));
                  __builtin_nontemporal_store(ou, reinterpret_cast<U4*>(reinterpret_cast<char*>(op) + cbytes[w]));
#else
                  *reinterpret_cast<VT*>(reinterpret_cast<char*>(op) + cbytes[w]) = o;
#endif
                }
#pragma unroll
                for (int x = 0; x < VEC; ++x) acc[w][x] = T(0);
              }
              ++j;
            }
          }
        }
      });
      if (last) {   // on to the group's next segment; the descriptor after it is fetched behind this chunk's prefetch
        j = jn;
        s0n = s0nn;
        remn = remnn;
        jn = jnn;
        seg += gstride;
        const bool more = seg < seg_hi;
        s0nn = more ? dx.start : 0;
        remnn = more ? dx.len : 0;
        jnn = more ? dx.info : 0;
      }
      rem = remq;
      cp = cq;
      vp = vq;
#pragma unroll
      for (int x = 0; x < EPL; ++x) {
        const bool in = li * EPL + x < rem;   // beyond the segment: index 0, value 0
        cl[x] = in ? cn[x] : 0u;
        vl[x] = in ? vn[x] : T(0);
        dl[x] = dn[x];
      }
    }
  }
  }  // tiles
}

// Sum of carry slots first + s0, first + s0 + stride, ... (< first + n) into acc, UB slots per batch: the loads of a
// batch are all issued before the first add (indices past the end are clamped to the last slot and their values
// dropped; lanes beyond the columns read column 0 and are never stored).  Written this way because hipcc, given the
// plain loop with its bounds tests, waits for every load before issuing the next one -- one 1 KiB load in flight per
// wave, 0.6 TB/s on the 660 MB of carries of C3.  The order of the additions is that of the plain loop.
template <typename T, int LPR, int VPL, int UB>
__device__ inline void sum_carry_slots(const T* __restrict__ carry, int64_t first, int n, int s0, int stride, int ld, int nc,
                                       int li, T (&acc)[VPL][VecOf<T>::N]) {
  constexpr int VEC = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  int cof[VPL];
#pragma unroll
  for (int w = 0; w < VPL; ++w) {
    const int co = (w * LPR + li) * VEC;
    cof[w] = co < nc ? co : 0;
  }
  for (int s = s0; s < n; s += stride * UB) {
    VT t[UB][VPL];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int su = s + u * stride;
      const T* cp = carry + (first + (su < n ? su : n - 1)) * ld;
#pragma unroll
      for (int w = 0; w < VPL; ++w) t[u][w] = *reinterpret_cast<const VT*>(cp + cof[w]);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const bool ok = s + u * stride < n;
#pragma unroll
      for (int w = 0; w < VPL; ++w)
#pragma unroll
        for (int x = 0; x < VEC; ++x) acc[w][x] += ok ? t[u][w].v[x] : T(0);
    }
  }
}

// One wave per chunked row: out[row] = sum of its carry slots in a fixed order.
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const int32_t* __restrict__ fix_row,
                                                         const int32_t* __restrict__ fix_first,
                                                         const int32_t* __restrict__ fix_count, int64_t n_fix,
                                                         const T* __restrict__ carry, int ld, int nc,
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
  sum_carry_slots<T, LPR, VPL, (VPL == 2 ? 4 : 8)>(carry, first, n, gi, NG, ld, nc, li, acc);
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
      if (co < nc) {
        VT o;
#pragma unroll
        for (int x = 0; x < VEC; ++x) o.v[x] = acc[w][x];
        *reinterpret_cast<VT*>(op + co) = o;
      }
    }
  }
}

// One block of NW waves per row with a long carry list (hot words own thousands of segments): the NW*NG lane
// groups stride the list, the groups of a wave are folded by shuffles and the NW waves through LDS in wave
// order -- a fixed summation tree, so the result does not depend on timing.
template <typename T, int LPR, int VPL, int NW>
__global__ __launch_bounds__(64 * NW) void spmm_fixup_long_kernel(const int32_t* __restrict__ fix_row,
                                                              const int32_t* __restrict__ fix_first,
                                                              const int32_t* __restrict__ fix_count,
                                                              const T* __restrict__ carry, int ld, int nc,
                                                              T* __restrict__ out) {
  constexpr int VEC = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  constexpr int NG = kWave / LPR;
  __shared__ T part[NW][LPR * VPL * VEC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const int64_t item = blockIdx.x;
  const int64_t first = fix_first[item];
  const int n = fix_count[item];
  T acc[VPL][VEC];
#pragma unroll
  for (int w = 0; w < VPL; ++w)
#pragma unroll
    for (int x = 0; x < VEC; ++x) acc[w][x] = T(0);
  sum_carry_slots<T, LPR, VPL, (VPL == 2 ? 4 : 8)>(carry, first, n, wv * NG + gi, NW * NG, ld, nc, li, acc);
#pragma unroll
  for (int mask = LPR; mask < kWave; mask <<= 1)
#pragma unroll
    for (int w = 0; w < VPL; ++w)
#pragma unroll
      for (int x = 0; x < VEC; ++x) acc[w][x] += shfl_xor_t(acc[w][x], mask);
  if (gi == 0) {
#pragma unroll
    for (int w = 0; w < VPL; ++w)
#pragma unroll
      for (int x = 0; x < VEC; ++x) part[wv][(w * LPR + li) * VEC + x] = acc[w][x];
  }
  __syncthreads();
  if (wv == 0 && gi == 0) {
    T* op = out + (int64_t)fix_row[item] * ld;
#pragma unroll
    for (int w = 0; w < VPL; ++w) {
      const int co = (w * LPR + li) * VEC;
      if (co < nc) {
        VT o;
#pragma unroll
        for (int x = 0; x < VEC; ++x) {
          T t = part[0][co + x];
#pragma unroll
          for (int w2 = 1; w2 < NW; ++w2) t += part[w2][co + x];
          o.v[x] = t;
        }
        *reinterpret_cast<VT*>(op + co) = o;
      }
    }
  }
}

// Both of the above in one launch of 256-thread blocks, for small streams where a launch costs as much as the work:
// blocks [0, n_long) take one long row each (the block-per-row tree with four waves), every later block four short
// rows, a wave each (the wave-per-row tree).  A row with a count of zero gets zeros: the rows without nnz ride along.
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void spmm_fixup_both_kernel(const int32_t* __restrict__ fix_row,
                                                              const int32_t* __restrict__ fix_first,
                                                              const int32_t* __restrict__ fix_count, int n_long,
                                                              int64_t n_fix, const T* __restrict__ carry, int ld, int nc,
                                                              T* __restrict__ out) {
  constexpr int VEC = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  constexpr int NG = kWave / LPR;
  constexpr int NW = 4;
  __shared__ T part[NW][LPR * VPL * VEC];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gi = lane / LPR, li = lane % LPR;
  const bool longrow = (int)blockIdx.x < n_long;   // block-uniform
  const int64_t item = longrow ? (int64_t)blockIdx.x : (int64_t)n_long + ((int64_t)blockIdx.x - n_long) * NW + wv;
  const bool active = item < n_fix;                // wave-uniform
  const int64_t first = active ? fix_first[item] : 0;
  const int n = active ? fix_count[item] : 0;
  T acc[VPL][VEC];
#pragma unroll
  for (int w = 0; w < VPL; ++w)
#pragma unroll
    for (int x = 0; x < VEC; ++x) acc[w][x] = T(0);
  sum_carry_slots<T, LPR, VPL, (VPL == 2 ? 4 : 8)>(carry, first, n, longrow ? wv * NG + gi : gi, longrow ? NW * NG : NG, ld, nc, li, acc);
#pragma unroll
  for (int mask = LPR; mask < kWave; mask <<= 1)
#pragma unroll
    for (int w = 0; w < VPL; ++w)
#pragma unroll
      for (int x = 0; x < VEC; ++x) acc[w][x] += shfl_xor_t(acc[w][x], mask);
  if (longrow) {
    if (gi == 0) {
#pragma unroll
      for (int w = 0; w < VPL; ++w)
#pragma unroll
        for (int x = 0; x < VEC; ++x) part[wv][(w * LPR + li) * VEC + x] = acc[w][x];
    }
    __syncthreads();
    if (wv != 0) return;
#pragma unroll
    for (int w = 0; w < VPL; ++w)
#pragma unroll
      for (int x = 0; x < VEC; ++x) {
        const int co = (w * LPR + li) * VEC + x;
        T t = part[0][co];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) t += part[w2][co];
        acc[w][x] = t;
      }
  }
  if (active && gi == 0) {
    T* op = out + (int64_t)fix_row[item] * ld;
#pragma unroll
    for (int w = 0; w < VPL; ++w) {
      const int co = (w * LPR + li) * VEC;
      if (co < nc) {
        VT o;
#pragma unroll
        for (int x = 0; x < VEC; ++x) o.v[x] = acc[w][x];
        *reinterpret_cast<VT*>(op + co) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Synchronised sweep: X^T.B for the RARE words (the ones the blocked stream leaves uncut: their nnz gather rows of B
// scattered over the whole table, one 256-byte row from the fabric each, at the ~29 G rows/s the fabric delivers in
// random order -- 0.8 ms of the 1.35 ms of X^T.W at C2).  Here every workgroup (one per CU, persistent) owns a slice of
// the rare vocabulary, with the output rows of its words as ACCUMULATORS IN LDS, and walks its nnz in document windows
// -- all workgroups window after window, in step (one barrier per window inside the workgroup, nothing between
// workgroups: equal work per window keeps them within a fraction of a window of each other).  The workgroups of an XCD
// then gather from the same few MB of B at the same time and a row fetched by one CU is an L2 hit for the others:
// tools/probes/sweep_probe.hip measures 65 G rows/s this way against 29 G rows/s in random order.  No carries, no
// fix-up; the sum of a word is formed in document order (windows ascending, documents ascending inside a window).
//
// Layout (sparse_prep / engine.hip:ensure_xt): nnz sorted by (window, partition, local row, document); cell = window *
// P + partition; each cell is cut into NGW = 16 * NG segments at word boundaries, one per lane group of the 1024-thread
// workgroup, so within a window a word is accumulated by exactly one lane group (read-modify-write of its LDS row, in
// order; deterministic) and the barrier orders the windows.  seg[cell * NGW + g] .. seg[cell * NGW + g + 1] delimit the
// segment of group g (contiguous over the whole stream, one sentinel at the end).
template <typename T>
struct SweepArgs {
  const int32_t* seg;        // [n_cells * NGW + 1] nnz offsets
  int n_windows;             // cells = n_windows * gridDim.x (partition = blockIdx.x)
  const uint32_t* docs;      // gather index (row of B)
  const uint32_t* lrows;     // local accumulator row of the nnz's word
  const T* vals;
  const T* B;                // tab_rows x ld
  int ld;
  int lds_ld;
  int rpp;                   // accumulator rows per partition
  uint32_t tab_bytes;
  const int32_t* words;      // [rpp * P] output row of accumulator row r of partition p at r * P + p, -1: none
  T* out;
};

template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(1024) void spmm_sweep_kernel(const SweepArgs<T> a) {
  constexpr int VEC = VecOf<T>::N;
  using VT = typename VecOf<T>::type;
  using U4 = __attribute__((ext_vector_type(4))) unsigned int;
  constexpr int NG = kWave / LPR;
  constexpr int NGW = 16 * NG;
  constexpr int EPL = (LPR >= 32) ? 1 : 32 / LPR;   // chunk = max(32, LPR) nnz: a segment is usually one chunk
  constexpr int CH = LPR * EPL;
  constexpr int U = VPL == 2 ? 4 : 8;
  extern __shared__ __align__(16) unsigned char mnmf_smem[];
  T* lds = reinterpret_cast<T*>(mnmf_smem);
  const int ld = a.ld;
  const int nvr = ld / VEC;
  for (int i = threadIdx.x; i < a.rpp * nvr; i += blockDim.x) {
    const int r = i / nvr, c = (i - r * nvr) * VEC;
    VT z;
#pragma unroll
    for (int x = 0; x < VEC; ++x) z.v[x] = T(0);
    *reinterpret_cast<VT*>(lds + (size_t)r * a.lds_ld + c) = z;
  }
  const int lane = threadIdx.x & 63;
  const int gi = lane / LPR, li = lane % LPR;
  const int g = (threadIdx.x >> 6) * NG + gi;
  bool colok[VPL];
  uint32_t cbytes[VPL];
#pragma unroll
  for (int w = 0; w < VPL; ++w) {
    const int coff = (w * LPR + li) * VEC;
    colok[w] = coff < ld;
    cbytes[w] = colok[w] ? (uint32_t)coff * (uint32_t)sizeof(T) : 0u;
  }
  const uint32_t rowbytes = (uint32_t)ld * (uint32_t)sizeof(T);
  const uint32_t lds_rowbytes = (uint32_t)a.lds_ld * (uint32_t)sizeof(T);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(a.B), 0, a.tab_bytes, 0x00020000);
  const int64_t P = gridDim.x;
  // this group's segment of window w: seg[(w * P + blockIdx.x) * NGW + g]
  auto seg_at = [&](int w) { return a.seg + ((int64_t)w * P + blockIdx.x) * NGW + g; };
  // the first chunk of a segment, fetched one window ahead (its bounds two windows ahead)
  uint32_t dn[EPL], rn[EPL];
  T vn[EPL];
  int s_next = 0, e_next = 0, s_next2 = 0, e_next2 = 0;
  auto fetch_chunk = [&](int s0, int len) {
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
      const int q = li * EPL + x;
      dn[x] = q < len ? a.docs[s0 + q] : 0u;
      rn[x] = q < len ? a.lrows[s0 + q] : 0u;
      vn[x] = q < len ? a.vals[s0 + q] : T(0);
    }
  };
  if (a.n_windows > 0) {
    s_next = seg_at(0)[0];
    e_next = seg_at(0)[1];
    fetch_chunk(s_next, e_next - s_next);
  }
  if (a.n_windows > 1) {
    s_next2 = seg_at(1)[0];
    e_next2 = seg_at(1)[1];
  }
  __syncthreads();   // accumulators are zero
  for (int w = 0; w < a.n_windows; ++w) {
    const int s0 = s_next, len = e_next - s_next;
    uint32_t dl[EPL], rl[EPL];
    T vl[EPL];
#pragma unroll
    for (int x = 0; x < EPL; ++x) {
      dl[x] = dn[x];
      rl[x] = rn[x];
      vl[x] = vn[x];
    }
    s_next = s_next2;
    e_next = e_next2;
    if (w + 1 < a.n_windows) fetch_chunk(s_next, e_next - s_next);
    if (w + 2 < a.n_windows) {
      s_next2 = seg_at(w + 2)[0];
      e_next2 = seg_at(w + 2)[1];
    }
    for (int base = 0; __any(base < len); base += CH) {
      if (base > 0) {   // a segment longer than one chunk (rare): fetched in place
#pragma unroll
        for (int x = 0; x < EPL; ++x) {
          const int q = base + li * EPL + x;
          dl[x] = q < len ? a.docs[s0 + q] : 0u;
          rl[x] = q < len ? a.lrows[s0 + q] : 0u;
          vl[x] = q < len ? a.vals[s0 + q] : T(0);
        }
      }
      static_for<CH / U>([&](auto ib) {
        constexpr int u0 = decltype(ib)::value * U;
        if (__any(base + u0 < len)) {   // wave-uniform: no group of this wave has nnz left in this batch
          uint32_t c[U], r[U];
          T v[U];
          VT b[U][VPL];
          static_for<U>([&](auto iu) {
            constexpr int u = decltype(iu)::value;
            constexpr int t = u0 + u;
            c[u] = (uint32_t)group_bcast_i<LPR, t / EPL>((int)dl[t % EPL], lane);
            r[u] = (uint32_t)group_bcast_i<LPR, t / EPL>((int)rl[t % EPL], lane);
            v[u] = group_bcast_t<LPR, t / EPL>(vl[t % EPL], lane);
          });
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int w2 = 0; w2 < VPL; ++w2) {
              const U4 gth = __builtin_amdgcn_raw_buffer_load_b128(rsrc, __umul24(c[u], rowbytes) + cbytes[w2], 0, 0);
              __builtin_memcpy(&b[u][w2], &gth, sizeof(VT));
            }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (base + u0 + u < len) {   // per lane group; the accumulator row belongs to this group in this window
              char* row = reinterpret_cast<char*>(lds) + __umul24(r[u], lds_rowbytes);
#pragma unroll
              for (int w2 = 0; w2 < VPL; ++w2) {
                if (colok[w2]) {
                  VT acc = *reinterpret_cast<VT*>(row + cbytes[w2]);
#pragma unroll
                  for (int x = 0; x < VEC; ++x) acc.v[x] += v[u] * b[u][w2].v[x];
                  *reinterpret_cast<VT*>(row + cbytes[w2]) = acc;
                }
              }
            }
          }
        }
      });
    }
    __syncthreads();   // the next window's owner of a row may be another lane group
  }
  // accumulators -> output rows
  for (int i = threadIdx.x; i < a.rpp * nvr; i += blockDim.x) {
    const int r = i / nvr, c = (i - r * nvr) * VEC;
    const int32_t word = a.words[(int64_t)r * P + blockIdx.x];
    if (word >= 0) *reinterpret_cast<VT*>(a.out + (int64_t)word * ld + c) = *reinterpret_cast<const VT*>(lds + (size_t)r * a.lds_ld + c);
  }
}

template <typename T>
__global__ void zero_rows_kernel(const int32_t* __restrict__ rows, int64_t n, int ld, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ld) return;
  out[(int64_t)rows[i / ld] * ld + (i % ld)] = T(0);
}

}  // namespace mnmf
