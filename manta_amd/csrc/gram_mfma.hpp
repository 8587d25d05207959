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
#include "dense.hpp"

#ifndef MNMF_RK_FAST_DIV
#define MNMF_RK_FAST_DIV 0
#endif
#ifndef MNMF_RK_PREFETCH
#define MNMF_RK_PREFETCH 1   // dense_rk_mfma_f32_kernel: epilogue operands fetched one group ahead (0: in front of their group)
#endif

namespace mnmf {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// SYM (A == B): only tiles on or above the diagonal are computed and the block writes their mirror images too
// (bit-identical to the full product: the same products in the same order).  With TR == NT a wave owns the whole
// upper triangle, the A operands are the B registers and nothing is loaded twice.
template <int NT, int TR, bool DIFF, bool SYM>
__global__ __launch_bounds__(512) void gram_mfma_f32_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                            const float* __restrict__ B, const float* __restrict__ B2,
                                                            int64_t rows, int ld, int64_t rpb,
                                                            float* __restrict__ slab) {
  constexpr int UR = DIFF ? 2 : 4;  // row pairs per batch; the next batch is in flight while this one multiplies
  constexpr int NW = 8;             // waves per block
  constexpr bool FULL = (TR == NT);
  constexpr bool AREG = SYM && FULL;  // A operands are B registers
  __shared__ float red[TR * NT * 16 * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cl = lane & 31, rh = lane >> 5;
  const int qi0 = FULL ? 0 : blockIdx.y * TR;
  f32x16 acc[TR][NT];
#pragma unroll
  for (int t = 0; t < TR; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][q][e] = 0.f;
  const int64_t r_begin = (int64_t)blockIdx.x * rpb;
  const int64_t r_end = (r_begin + rpb < rows) ? r_begin + rpb : rows;
  if (r_begin >= r_end) return;  // never happens: the host sizes the grid to cdiv(rows, rpb)
  bool aok[TR], bok[NT];
  int acol[TR], bcol[NT];
#pragma unroll
  for (int t = 0; t < TR; ++t) {
    const int c = 32 * (qi0 + t) + cl;
    aok[t] = c < ld;
    acol[t] = aok[t] ? c : ld - 1;
  }
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int c = 32 * q + cl;
    bok[q] = c < ld;
    bcol[q] = bok[q] ? c : ld - 1;
  }
  constexpr int NA = AREG ? 1 : TR;
  float an[UR][NA][DIFF ? 2 : 1], bn[UR][NT][DIFF ? 2 : 1];
  // branch-free fetch: rows past the end and columns past ld read a clamped address and are zeroed at use
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      int64_t row = r0 + 2 * NW * u + rh;
      row = row < r_end ? row : r_end - 1;
      const int64_t off = row * ld;
      if (!AREG) {
#pragma unroll
        for (int t = 0; t < TR; ++t) {
          an[u][t][0] = A[off + acol[t]];
          if (DIFF) an[u][t][1] = A2[off + acol[t]];
        }
      }
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        if (SYM && !FULL && q < qi0) continue;  // block-uniform: tiles left of the diagonal are not computed
        bn[u][q][0] = B[off + bcol[q]];
        if (DIFF) bn[u][q][1] = B2[off + bcol[q]];
      }
    }
  };
  const int64_t stride = 2 * NW * UR;
  int64_t r0 = r_begin + 2 * wv;
  if (r0 < r_end) fetch(r0);
  for (; r0 < r_end; r0 += stride) {
    float a[UR][TR], b[UR][NT];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const bool rok = r0 + 2 * NW * u + rh < r_end;
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        float x = bn[u][q][0];
        if (DIFF) x = fabsf(x - bn[u][q][1]);
        b[u][q] = (rok && bok[q]) ? x : 0.f;
      }
#pragma unroll
      for (int t = 0; t < TR; ++t) {
        if (AREG) {
          a[u][t] = b[u][t];
        } else {
          float x = an[u][t][0];
          if (DIFF) x = fabsf(x - an[u][t][1]);
          a[u][t] = (rok && aok[t]) ? x : 0.f;
        }
      }
    }
    if (r0 + stride < r_end) fetch(r0 + stride);
#pragma unroll
    for (int u = 0; u < UR; ++u)
#pragma unroll
      for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          if (SYM && q < qi0 + t) continue;
          acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][q], acc[t][q], 0, 0, 0);
        }
  }
  // block reduction in wave order through LDS (element (t,q,e) of lane l at ((t*NT+q)*16+e)*64 + l)
  for (int w = 0; w < NW; ++w) {
    if (wv == w) {
#pragma unroll
      for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          if (SYM && q < qi0 + t) continue;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int idx = ((t * NT + q) * 16 + e) * 64 + lane;
            red[idx] = (w == 0) ? acc[t][q][e] : red[idx] + acc[t][q][e];
          }
        }
    }
    __syncthreads();
  }
  // C/D layout of the 32x32 tile: lane holds column l&31, rows (e&3) + 8*(e>>2) + 4*(l>>5)
  float* dst = slab + (int64_t)blockIdx.x * (int64_t)ld * ld;
  for (int i = threadIdx.x; i < TR * NT * 16 * 64; i += NW * 64) {
    const int l = i & 63, e = (i >> 6) & 15, tq = i >> 10;
    const int t = tq / NT, q = tq % NT;
    if (SYM && q < qi0 + t) continue;
    const int gi = 32 * (qi0 + t) + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
    const int gj = 32 * q + (l & 31);
    if (gi < ld && gj < ld) {
      const float v = red[i];
      dst[(int64_t)gi * ld + gj] = v;
      if (SYM && q > qi0 + t) dst[(int64_t)gj * ld + gi] = v;
    }
  }
}

// Wide symmetric Gram (129 <= ld <= 256, NT = 5..8 column tiles).  With one tile row per workgroup (the kernel above
// with TR = 1) the factor is read NT/2 + 1 times from HBM -- 10.8 GB instead of 2 GB at k = 200 on 2.5M rows, which is
// what bounded it.  Here a workgroup owns a row range and its 8 waves split the TILES of the upper triangle (3..5
// each, contiguous in row-major order): every wave walks all rows of the range, the operand columns it needs come
// from L1 after the first wave touched them, and the factor leaves HBM once.  The tile set of a wave is a compile-time
// constant (one instantiation of the body per wave index), so each distinct operand column is loaded once per row
// pair however many of the wave's tiles use it -- about one load per MFMA instead of two, which matters because this
// kernel is bound by the vector-memory issue rate.  A wave keeps its tiles in registers for the whole range and
// writes them (and their mirror images) straight to the block's slab partial: no cross-wave reduction, no LDS.
// Same fixed summation order on every run.
template <int NT>
struct TriTile {  // tile idx (row-major over the upper triangle) -> (t, q)
  static constexpr int row(int idx) {
    int t = 0;
    while (idx >= NT - t) {
      idx -= NT - t;
      ++t;
    }
    return t;
  }
  static constexpr int col(int idx) {
    int t = 0;
    while (idx >= NT - t) {
      idx -= NT - t;
      ++t;
    }
    return t + idx;
  }
};

template <int NT, bool DIFF, int WV>
__device__ inline void gram_wide_body(const float* __restrict__ A, const float* __restrict__ A2, int ld, int64_t r_begin,
                                      int64_t r_end, float* __restrict__ dst) {
  constexpr int NTRI = NT * (NT + 1) / 2;
  constexpr int LO = WV * NTRI / 8, HI = (WV + 1) * NTRI / 8, N = HI - LO;
  constexpr int UR = DIFF ? 2 : 4;  // row pairs per batch
  if constexpr (N > 0) {
    const int lane = threadIdx.x & 63;
    const int cl = lane & 31, rh = lane >> 5;
    // which operand columns this wave needs (compile time)
    auto needed = [](int c) constexpr {
      for (int j = 0; j < N; ++j)
        if (TriTile<NT>::row(LO + j) == c || TriTile<NT>::col(LO + j) == c) return true;
      return false;
    };
    int ccol[NT];
    bool cok[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
      const int x = 32 * c + cl;
      cok[c] = x < ld;
      ccol[c] = cok[c] ? x : ld - 1;
    }
    f32x16 acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float vn[UR][NT][DIFF ? 2 : 1];
    auto fetch = [&](int64_t r0) {
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        int64_t row = r0 + 2 * u + rh;
        row = row < r_end ? row : r_end - 1;
        const int64_t off = row * ld;
        static_for<NT>([&](auto c) {
          if constexpr (needed(c.value)) {
            vn[u][c.value][0] = A[off + ccol[c.value]];
            if (DIFF) vn[u][c.value][1] = A2[off + ccol[c.value]];
          }
        });
      }
    };
    constexpr int64_t stride = 2 * UR;
    fetch(r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += stride) {
      float v[UR][NT];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const bool rok = r0 + 2 * u + rh < r_end;
        static_for<NT>([&](auto c) {
          if constexpr (needed(c.value)) {
            float x = vn[u][c.value][0];
            if (DIFF) x = fabsf(x - vn[u][c.value][1]);
            v[u][c.value] = (rok && cok[c.value]) ? x : 0.f;
          }
        });
      }
      if (r0 + stride < r_end) fetch(r0 + stride);
#pragma unroll
      for (int u = 0; u < UR; ++u)
        static_for<N>([&](auto j) {
          constexpr int t = TriTile<NT>::row(LO + j.value), q = TriTile<NT>::col(LO + j.value);
          acc[j.value] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u][t], v[u][q], acc[j.value], 0, 0, 0);
        });
    }
    // C/D layout of the 32x32 tile: lane holds column l&31, rows (e&3) + 8*(e>>2) + 4*(l>>5)
    static_for<N>([&](auto j) {
      constexpr int t = TriTile<NT>::row(LO + j.value), q = TriTile<NT>::col(LO + j.value);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int gi = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * rh;
        const int gj = 32 * q + cl;
        if (gi < ld && gj < ld) {
          const float x = acc[j.value][e];
          dst[(int64_t)gi * ld + gj] = x;
          if (q > t) dst[(int64_t)gj * ld + gi] = x;
        }
      }
    });
  }
}

template <int NT, bool DIFF>
__global__ __launch_bounds__(512) void gram_mfma_wide_f32_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                                                 int64_t rows, int ld, int64_t rpb, float* __restrict__ slab) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t r_begin = (int64_t)blockIdx.x * rpb;
  const int64_t r_end = (r_begin + rpb < rows) ? r_begin + rpb : rows;
  if (r_begin >= r_end) return;  // never happens: the host sizes the grid to cdiv(rows, rpb)
  float* dst = slab + (int64_t)blockIdx.x * (int64_t)ld * ld;
  switch (wv) {  // one body per wave index: its tile set is a compile-time constant
    case 0: gram_wide_body<NT, DIFF, 0>(A, A2, ld, r_begin, r_end, dst); break;
    case 1: gram_wide_body<NT, DIFF, 1>(A, A2, ld, r_begin, r_end, dst); break;
    case 2: gram_wide_body<NT, DIFF, 2>(A, A2, ld, r_begin, r_end, dst); break;
    case 3: gram_wide_body<NT, DIFF, 3>(A, A2, ld, r_begin, r_end, dst); break;
    case 4: gram_wide_body<NT, DIFF, 4>(A, A2, ld, r_begin, r_end, dst); break;
    case 5: gram_wide_body<NT, DIFF, 5>(A, A2, ld, r_begin, r_end, dst); break;
    case 6: gram_wide_body<NT, DIFF, 6>(A, A2, ld, r_begin, r_end, dst); break;
    default: gram_wide_body<NT, DIFF, 7>(A, A2, ld, r_begin, r_end, dst); break;
  }
}

// ------------------------------------------------------------------------------------------------
// rows x (k x k) products fused with the multiplicative update, on the matrix cores (fp32):
//   acc = A1.G1 (+ A2.G2);  PROD: out = acc;  MU: out = upd*(num/(acc+eps));  MU_SQRT: upd*sqrt(num/(acc+eps))
// (nmf_basic.py:41-42,46-47; nmf_projective_basic.py:76-79; nmtf.py:151-153,161-163).
// One wave owns RT*32 rows x all NT*32 columns.  The 32x32x2 MFMA wants lane (i, h) to hold A[row i][k + h];
// any pairing of the two k's works as long as G's rows are paired the same way, so lane (i, h) loads the
// 32 contiguous bytes A[row i][k0 + 8h .. k0 + 8h + 7] (two 16-byte loads) and step s of the chunk multiplies
// k = k0 + s (half 0) with k = k0 + 8 + s (half 1): no LDS, no barriers, every load independent of the MFMAs.
// G (ld x ld, cache resident) is read as coalesced 128-byte rows.  The next chunk's operands are fetched while
// the current chunk's MFMAs run.  Out-of-range k / rows / columns are clamped (A is zeroed there), never branched.
// The epilogue works in the accumulator layout (lane = column, two 128-byte row segments per access).
// FUSE (NT <= 2 only): the Grams that the solvers need right after an update are accumulated in the same pass,
// from the registers of the epilogue: bit 0 = (new)^T (new)  (the next iteration's w.T@w / h@h.T,
// nmf_basic.py:39,44), bit 1 = D^T D with D = |new - old| (the convergence norms, :52-53).  In the accumulator layout
// lane (c, h) holds rows rho + 4h of column c, which is a legal 32x32x2 operand pair (rows rho and rho + 4), so the
// epilogue's values feed the MFMA directly.  A wave then owns several row tiles (grid-stride) and keeps the Gram
// tiles in registers; blocks reduce through LDS in wave order and write one partial per block to the slabs.
// Shapes: ld % 16 == 0 for NT >= 2 (the engine pads k > 32 that way); NT == 1 takes any ld % 4 == 0 (TAIL selects).
// Memory access is through raw buffer descriptors rebuilt per tile (scalar work): the descriptor's size is the tile's
// valid bytes, so rows past the end read 0 and their stores are dropped by the hardware bounds check, and columns
// past ld (and, for NT == 1, k past ld) get a poisoned offset instead of a branch or a select.  Per-lane state is a
// handful of 32-bit offsets; everything tile-dependent is scalar.
template <int NT, int MODE, int FUSE>
__global__ __launch_bounds__(256, (NT <= 2 && FUSE == 0) ? 3 : (NT <= 4 ? 2 : 1))  // min waves per SIMD
void dense_rk_mfma_f32_kernel(RkArgs<float> a, int64_t n_tiles, float* __restrict__ slab_new,
                              float* __restrict__ slab_diff) {
  static_assert(FUSE == 0 || (NT <= 2 && MODE != RK_PROD), "fused Grams: k <= 64");
  using U4 = __attribute__((ext_vector_type(4))) unsigned int;
  constexpr bool TAIL = (NT == 1);
  constexpr int NTRI = NT * (NT + 1) / 2;
  constexpr unsigned kPoison = 0x40000000u;  // beyond any descriptor size used here
  constexpr int kFlags = 0x00020000;
  __shared__ float red[FUSE ? NTRI * 16 * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: tile bases stay scalar
  const int cl = lane & 31, rh = lane >> 5;
  const int ld = a.ld;
  const unsigned rowb = (unsigned)ld * 4u;
  const int kc = a.kcols > 0 ? a.kcols : ld;   // columns of this call (batched sweep: one problem's; RkArgs)
  const unsigned cob = 4u * (unsigned)a.coff;  // ... and where they start inside the padded row, in bytes
  bool cok[NT];
  unsigned goff[NT], eoff[NT];  // byte offsets: G operand (row 8*rh, column), epilogue (row 4*rh, column)
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int col = 32 * NT * (int)blockIdx.y + 32 * q + cl;   // blockIdx.y: column group of 32*NT columns (ld > 128)
    cok[q] = col < kc;
    goff[q] = cok[q] ? (unsigned)(8 * rh) * rowb + 4u * col : kPoison;
    eoff[q] = cok[q] ? (unsigned)(4 * rh) * rowb + 4u * col + cob : kPoison;
  }
  const unsigned aoff = (unsigned)cl * rowb + 32u * rh + cob;  // A operand: row cl, 8 floats at column 8*rh
  f32x16 gnew[(FUSE & 1) ? NTRI : 1], gdif[(FUSE & 2) ? NTRI : 1];
  if (FUSE) {
#pragma unroll
    for (int x = 0; x < NTRI; ++x)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (FUSE & 1) gnew[x][e] = 0.f;
        if (FUSE & 2) gdif[x][e] = 0.f;
      }
  }

#pragma unroll 1
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t row0 = tile * 32;
    const int64_t tbase = row0 * ld;
    const int rleft = (int)((a.rows - row0 < 32) ? a.rows - row0 : 32);   // valid rows in this tile
    const unsigned tbytes = (unsigned)rleft * rowb;
    f32x16 acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
    // The epilogue works in groups of EG accumulator rows.  Numerator and old factor of a group are fetched one group
    // ahead -- the first group's before the products below, the next group's before this group's stores (the compiler
    // cannot prove that `out` does not alias them and never moves a load above a store) -- so their trip to memory
    // runs behind MFMAs instead of in front of them (the matrix cores idled half of the kernel).
    constexpr int EG = NT <= 2 ? 8 : 4;
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.num + (MODE != RK_PROD ? tbase : 0)), 0, tbytes, kFlags);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.upd + (MODE != RK_PROD ? tbase : 0)), 0, tbytes, kFlags);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + tbase, 0, tbytes, kFlags);
    float nv[2][NT][EG], uv[2][NT][EG];
    auto load_group = [&](auto bufc, int e0) {
      constexpr int buf = decltype(bufc)::value;
      if (MODE != RK_PROD) {
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int x = 0; x < EG; ++x) {
            const int e = e0 + x;
            const unsigned off = eoff[q] + (unsigned)((e & 3) + 8 * (e >> 2)) * rowb;
            nv[buf][q][x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rn, off, 0, 0));
            uv[buf][q][x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, off, 0, 0));
          }
      }
    };
    constexpr bool PF = MNMF_RK_PREFETCH && NT <= 4;   // wider tiles have no registers to spare
    if (PF) load_group(std::integral_constant<int, 0>{}, 0);
#pragma unroll 1
    for (int pair = 0; pair < 2; ++pair) {
      const float* A = pair == 0 ? a.A1 : a.A2;
      const float* G = pair == 0 ? a.G1 : a.G2;
      if (A == nullptr) break;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + tbase), 0, tbytes, kFlags);
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, (unsigned)ld * rowb, kFlags);
      U4 an[2];
      unsigned gn[8][NT];
      auto fetch = [&](int k0) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          if (TAIL) {
            const int c = k0 + 8 * rh + 4 * x;
            an[x] = __builtin_amdgcn_raw_buffer_load_b128(ra, c < ld ? aoff + 4u * (k0 + 4 * x) : kPoison, 0, 0);
          } else {
            an[x] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff + 16u * x, 4 * k0, 0);
          }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            if (TAIL) gn[s][q] = __builtin_amdgcn_raw_buffer_load_b32(rg, goff[q] + (unsigned)(k0 + s) * rowb, 0, 0);
            else gn[s][q] = __builtin_amdgcn_raw_buffer_load_b32(rg, goff[q], (k0 + s) * (int)rowb, 0);
          }
      };
      fetch(0);
#pragma unroll 1
      for (int k0 = 0; k0 < ld; k0 += 16) {
        U4 ac[2];
        unsigned gc[8][NT];
        ac[0] = an[0];
        ac[1] = an[1];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int q = 0; q < NT; ++q) gc[s][q] = gn[s][q];
        if (k0 + 16 < ld) fetch(k0 + 16);
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int q = 0; q < NT; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(ac[s >> 2][s & 3]), __uint_as_float(gc[s][q]),
                                                          acc[q], 0, 0, 0);
      }
    }
    // epilogue in groups of EG accumulator rows: a group's loads are all issued before the previous group's first
    // store; with FUSE the group's new values and differences go straight into the Gram MFMAs.
    static_for<16 / EG>([&](auto gc_) {
      constexpr int grp = decltype(gc_)::value;
      constexpr int e0 = grp * EG;
      constexpr int cur = PF ? (grp & 1) : 0;
      if (!PF) load_group(std::integral_constant<int, 0>{}, e0);
      else if (grp + 1 < 16 / EG) load_group(std::integral_constant<int, 1 - cur>{}, e0 + EG);
      float on[NT][EG], od[NT][EG];
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int x = 0; x < EG; ++x) {
          const int e = e0 + x;
          const int r = 4 * rh + (e & 3) + 8 * (e >> 2);
          const unsigned off = eoff[q] + (unsigned)((e & 3) + 8 * (e >> 2)) * rowb;
          float o = acc[q][e];
          if (MODE != RK_PROD) {
#if MNMF_RK_FAST_DIV   // timing experiment: 1-ulp reciprocal instead of the correctly rounded quotient
            const float qn = nv[cur][q][x] * __builtin_amdgcn_rcpf(o + a.eps);
#else
            const float qn = nv[cur][q][x] / (o + a.eps);
#endif
            o = uv[cur][q][x] * (MODE == RK_MU_SQRT ? sqrtf(qn) : qn);
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), ro, off, 0, 0);   // dropped when out of the tile
          if (FUSE) {
            const bool ok = cok[q] && r < rleft;
            on[q][x] = ok ? o : 0.f;
            od[q][x] = ok ? fabsf(o - uv[cur][q][x]) : 0.f;
          }
        }
      if (FUSE) {
#pragma unroll
        for (int x = 0; x < EG; ++x) {
          int y = 0;
#pragma unroll
          for (int qa = 0; qa < NT; ++qa)
#pragma unroll
            for (int qb = qa; qb < NT; ++qb, ++y) {
              if (FUSE & 1) gnew[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(on[qa][x], on[qb][x], gnew[y], 0, 0, 0);
              if (FUSE & 2) gdif[y] = __builtin_amdgcn_mfma_f32_32x32x2f32(od[qa][x], od[qb][x], gdif[y], 0, 0, 0);
            }
        }
      }
    });
  }
  if (FUSE) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      if (!(FUSE & (1 << which))) continue;
      __syncthreads();  // red is reused
      for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
          for (int x = 0; x < NTRI; ++x)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int idx = (x * 16 + e) * 64 + lane;
              const float v = which == 0 ? gnew[x][e] : gdif[x][e];
              red[idx] = (w == 0) ? v : red[idx] + v;
            }
        }
        __syncthreads();
      }
      float* dst = (which == 0 ? slab_new : slab_diff) + (int64_t)blockIdx.x * (int64_t)ld * ld;
      for (int i = threadIdx.x; i < NTRI * 16 * 64; i += 256) {
        const int l = i & 63, e = (i >> 6) & 15, x = i >> 10;
        int qa = 0, qb = x;  // x -> (qa, qb), qa <= qb, row-major over the upper triangle
        while (qb >= NT - qa) { qb -= NT - qa; ++qa; }
        qb += qa;
        const int gi = 32 * qa + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        const int gj = 32 * qb + (l & 31);
        if (gi < ld && gj < ld) {
          const float v = red[i];
          dst[(int64_t)gi * ld + gj] = v;
          if (qb > qa) dst[(int64_t)gj * ld + gi] = v;
        }
      }
    }
  }
}

}  // namespace mnmf
