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
template <int NT, int RT, int MODE>
__global__ __launch_bounds__(256) void dense_rk_mfma_f32_kernel(RkArgs<float> a) {
  using V4 = Vec<float, 4>;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cl = lane & 31, rh = lane >> 5;
  const int ld = a.ld;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wv) * (32 * RT);
  if (row0 >= a.rows) return;
  f32x16 acc[RT][NT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][q][e] = 0.f;
  int64_t arow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    int64_t r = row0 + 32 * t + cl;
    arow[t] = (r < a.rows ? r : a.rows - 1) * ld;
  }
  int gcol[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) gcol[q] = (32 * q + cl < ld) ? 32 * q + cl : ld - 1;

  for (int pair = 0; pair < 2; ++pair) {
    const float* A = pair == 0 ? a.A1 : a.A2;
    const float* G = pair == 0 ? a.G1 : a.G2;
    if (A == nullptr) break;
    V4 an[RT][2];
    float gn[8][NT];
    auto fetch = [&](int k0) {
      const int kc = k0 + 8 * rh;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int c = kc + 4 * x;
        const bool ok = c < ld;
        const int cc = ok ? c : ld - 4;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          V4 v = *reinterpret_cast<const V4*>(A + arow[t] + cc);
          if (!ok) v.v[0] = v.v[1] = v.v[2] = v.v[3] = 0.f;
          an[t][x] = v;
        }
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int kr = kc + s < ld ? kc + s : ld - 1;
        const float* gp = G + (int64_t)kr * ld;
#pragma unroll
        for (int q = 0; q < NT; ++q) gn[s][q] = gp[gcol[q]];
      }
    };
    fetch(0);
    for (int k0 = 0; k0 < ld; k0 += 16) {
      V4 ac[RT][2];
      float gc[8][NT];
#pragma unroll
      for (int t = 0; t < RT; ++t) { ac[t][0] = an[t][0]; ac[t][1] = an[t][1]; }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int q = 0; q < NT; ++q) gc[s][q] = gn[s][q];
      if (k0 + 16 < ld) fetch(k0 + 16);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int t = 0; t < RT; ++t)
            acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[t][s >> 2].v[s & 3], gc[s][q], acc[t][q], 0, 0, 0);
    }
  }
  // epilogue: all of a tile's loads are issued before its first store (the compiler cannot prove that out does not
  // alias num/upd, and would otherwise wait for memory once per element)
  const float* __restrict__ num = a.num;
  const float* __restrict__ upd = a.upd;
  float* __restrict__ out = a.out;
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int col = 32 * q + cl;
      const bool cok = col < ld;
      const int64_t rbase = row0 + 32 * t + 4 * rh;
      float nv[16], uv[16];
      if (MODE != RK_PROD) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int64_t row = rbase + (e & 3) + 8 * (e >> 2);
          const bool ok = cok && row < a.rows;
          const int64_t off = ok ? row * ld + col : 0;
          nv[e] = num[off];
          uv[e] = upd[off];
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = rbase + (e & 3) + 8 * (e >> 2);
        float o = acc[t][q][e];
        if (MODE != RK_PROD) {
          const float qn = nv[e] / (o + a.eps);
          o = uv[e] * (MODE == RK_MU_SQRT ? sqrtf(qn) : qn);
        }
        if (cok && row < a.rows) out[row * ld + col] = o;
      }
    }
}

}  // namespace mnmf
