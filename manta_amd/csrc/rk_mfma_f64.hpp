// fp64 on the matrix cores: rows x (k x k) products fused with the multiplicative update and, in the same pass, the
// Grams the solvers need right after an update (the fp64 counterpart of dense_rk_mfma_f32_kernel, gram_mfma.hpp):
//   acc = A1.G1 (+ A2.G2);  PROD: out = acc;  MU: out = upd*(num/(acc+eps));  MU_SQRT: upd*sqrt(num/(acc+eps))
// (nmf_basic.py:41-42,46-47; nmf_projective_basic.py:76-79; nmtf.py:151-153,161-163), FUSE bit 0: (new)^T (new)
// (nmf_basic.py:39,44), bit 1: D^T D with D = |new - old| (:52-53).
//
// v_mfma_f64_16x16x4_f64: D(16x16) += A(16x4).B(4x16); lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; the
// result has column l & 15 on the lane and rows (l >> 4) + 4*reg in its 4 registers (NOT the f32 maps).
// One wave owns 16 rows x all NT*16 columns.  As in the f32 kernel any assignment of k's to the 4 lane quarters is
// legal if G's rows follow it: lane (i, q) loads the 32 contiguous bytes A[row i][16c + 4q .. 16c + 4q + 3] and step s
// multiplies k = 16c + 4q + s -- no LDS, every load independent of the MFMAs.  In the result layout lane (j, q) holds
// rows q + 4*reg of column j: for a fixed reg that is exactly a legal operand (rows 4*reg .. 4*reg + 3 on the four
// quarters), so the epilogue's new values and |new - old| feed the Gram MFMAs directly.  Blocks reduce their Gram
// tiles through LDS in wave order and write one partial per block to the slabs; slab_reduce_kernel sums them in a
// fixed order.  The matrix cores run fp64 at the vector rate on this chip: the gain is one pass over the factor
// instead of three and no LDS staging, not flops.  ld in {16, 32, 48, 64} (what the engine pads k <= 64 to in fp64).
#pragma once
#include "common.hpp"
#include "dense.hpp"

namespace mnmf {

using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int NT, int MODE, int FUSE>
__global__ __launch_bounds__(256, 1) void dense_rk_mfma_f64_kernel(RkArgs<double> a, int64_t n_tiles,
                                                                   double* __restrict__ slab_new,
                                                                   double* __restrict__ slab_diff) {
  static_assert(FUSE == 0 || MODE != RK_PROD, "fused Grams belong to an update");
  using U4 = __attribute__((ext_vector_type(4))) unsigned int;
  using U2 = __attribute__((ext_vector_type(2))) unsigned int;
  constexpr int NTRI = NT * (NT + 1) / 2;
  constexpr int kFlags = 0x00020000;
  __shared__ double red[FUSE ? NTRI * 4 * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int ld = a.ld;
  const unsigned rowb = (unsigned)ld * 8u;
  // Batched topic-count sweep (RkArgs::coff / kcols, dense.hpp): this call works on the kcols columns that start at
  // column coff of the padded rows; G1 is the problem's own ld x ld matrix, zero beyond kcols.  The A operand is read
  // at the column offset (8-byte aligned 16-byte buffer loads; what it reads beyond the problem's columns meets zero
  // rows of G), numerator / old factor / output only in the problem's columns (a poisoned offset beyond any
  // descriptor size reads 0 and drops the store).
  constexpr unsigned kPoison = 0x40000000u;
  const int kc = a.kcols > 0 ? a.kcols : ld;
  const unsigned cob = 8u * (unsigned)a.coff;
  const unsigned aoff = (unsigned)j * rowb + 32u * q + cob;      // A operand: row j, 4 doubles at column 4q (+ 16c)
  unsigned goff[NT], eoff[NT];                                   // G operand: row 4q (+16c + s), column 16t + j
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    goff[t] = (unsigned)(4 * q) * rowb + 8u * (16 * t + j);
    eoff[t] = (16 * t + j < kc) ? (unsigned)q * rowb + 8u * (16 * t + j) + cob : kPoison;   // epilogue: row q (+ 4*reg), column 16t + j
  }
  f64x4 gnew[(FUSE & 1) ? NTRI : 1], gdif[(FUSE & 2) ? NTRI : 1];
  if (FUSE) {
#pragma unroll
    for (int x = 0; x < NTRI; ++x)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (FUSE & 1) gnew[x][e] = 0.0;
        if (FUSE & 2) gdif[x][e] = 0.0;
      }
  }
  auto as_double = [](unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); };

  // One wave per SIMD (the Gram accumulators fill the register file), so latency is hidden inside the wave: the
  // operands of the next 16-k chunk -- across the A1/A2 pairs and across row tiles -- are fetched while the current
  // chunk multiplies, and the epilogue's numerator / old-factor loads are issued behind the first chunk's operands
  // have arrived, a whole product ahead of their use.
  const int npairs = a.A2 != nullptr ? 2 : 1;
  const int64_t tstride = (int64_t)gridDim.x * 4;
  U4 an[2];
  U2 gn[4][NT];
  auto fetch = [&](int64_t tl, int pair, int c) {
    const double* A = pair == 0 ? a.A1 : a.A2;
    const double* G = pair == 0 ? a.G1 : a.G2;
    const int64_t r0 = tl * 16;
    const int left = (int)((a.rows - r0 < 16) ? a.rows - r0 : 16);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A + r0 * ld), 0, (unsigned)left * rowb, kFlags);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(G), 0, (unsigned)ld * rowb, kFlags);
    an[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff, 128 * c, 0);
    an[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, aoff + 16u, 128 * c, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < NT; ++t) gn[s][t] = __builtin_amdgcn_raw_buffer_load_b64(rg, goff[t], (16 * c + s) * (int)rowb, 0);
  };
  int64_t tile = (int64_t)blockIdx.x * 4 + wv;
  if (tile < n_tiles) fetch(tile, 0, 0);
#pragma unroll 1
  for (; tile < n_tiles; tile += tstride) {
    const int64_t row0 = tile * 16;
    const int64_t tbase = row0 * ld;
    const int rleft = (int)((a.rows - row0 < 16) ? a.rows - row0 : 16);
    const unsigned tbytes = (unsigned)rleft * rowb;   // rows past the end read 0, their stores are dropped
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.num + (MODE != RK_PROD ? tbase : 0)), 0, tbytes, kFlags);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.upd + (MODE != RK_PROD ? tbase : 0)), 0, tbytes, kFlags);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out + tbase, 0, tbytes, kFlags);
    f64x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][e] = 0.0;
    U2 nv2[NT][4], uv2[NT][4];
#pragma unroll 1
    for (int step = 0; step < npairs * NT; ++step) {   // ld = 16 * NT: chunks of 16 k's, pair after pair
      const int pair = step / NT, c = step - pair * NT;
      U4 ac[2] = {an[0], an[1]};
      U2 gc[4][NT];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t) gc[s][t] = gn[s][t];
      if (step == 0 && MODE != RK_PROD) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned off = eoff[t] + (unsigned)(4 * e) * rowb;
            nv2[t][e] = __builtin_amdgcn_raw_buffer_load_b64(rn, off, 0, 0);
            uv2[t][e] = __builtin_amdgcn_raw_buffer_load_b64(ru, off, 0, 0);
          }
      }
      if (c + 1 < NT) fetch(tile, pair, c + 1);
      else if (pair + 1 < npairs) fetch(tile, pair + 1, 0);
      else if (tile + tstride < n_tiles) fetch(tile + tstride, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double av = as_double(ac[s >> 1][2 * (s & 1)], ac[s >> 1][2 * (s & 1) + 1]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, as_double(gc[s][t][0], gc[s][t][1]), acc[t], 0, 0, 0);
      }
    }
    double nv[NT][4], uv[NT][4];
    if (MODE != RK_PROD) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          nv[t][e] = as_double(nv2[t][e][0], nv2[t][e][1]);
          uv[t][e] = as_double(uv2[t][e][0], uv2[t][e][1]);
        }
    }
    double on[NT][4], od[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned off = eoff[t] + (unsigned)(4 * e) * rowb;
        double o = acc[t][e];
        if (MODE != RK_PROD) {
          const double qn = nv[t][e] / (o + a.eps);
          o = uv[t][e] * (MODE == RK_MU_SQRT ? sqrt(qn) : qn);
        }
        U2 o2;
        o2[0] = (unsigned)__double2loint(o);
        o2[1] = (unsigned)__double2hiint(o);
        __builtin_amdgcn_raw_buffer_store_b64(o2, ro, off, 0, 0);
        if (FUSE) {
          const bool ok = q + 4 * e < rleft;
          on[t][e] = ok ? o : 0.0;
          od[t][e] = ok ? fabs(o - uv[t][e]) : 0.0;
        }
      }
    if (FUSE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int y = 0;
#pragma unroll
        for (int ta = 0; ta < NT; ++ta)
#pragma unroll
          for (int tb = ta; tb < NT; ++tb, ++y) {
            if (FUSE & 1) gnew[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(on[ta][e], on[tb][e], gnew[y], 0, 0, 0);
            if (FUSE & 2) gdif[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(od[ta][e], od[tb][e], gdif[y], 0, 0, 0);
          }
      }
    }
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
            for (int e = 0; e < 4; ++e) {
              const int idx = (x * 4 + e) * 64 + lane;
              const double v = which == 0 ? gnew[x][e] : gdif[x][e];
              red[idx] = (w == 0) ? v : red[idx] + v;
            }
        }
        __syncthreads();
      }
      double* dst = (which == 0 ? slab_new : slab_diff) + (int64_t)blockIdx.x * (int64_t)ld * ld;
      for (int i = threadIdx.x; i < NTRI * 4 * 64; i += 256) {
        const int l = i & 63, e = (i >> 6) & 3, x = i >> 8;
        int ta = 0, tb = x;  // x -> (ta, tb), ta <= tb, row-major over the upper triangle
        while (tb >= NT - ta) { tb -= NT - ta; ++ta; }
        tb += ta;
        const int gi = 16 * ta + (l >> 4) + 4 * e;   // f64 result map: rows (l >> 4) + 4*reg
        const int gj = 16 * tb + (l & 15);
        const double v = red[i];
        dst[(int64_t)gi * ld + gj] = v;
        if (tb > ta) dst[(int64_t)gj * ld + gi] = v;
      }
    }
  }
}

}  // namespace mnmf
