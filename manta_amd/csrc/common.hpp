// Shared helpers for the mnmf engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <type_traits>
#include <utility>

namespace mnmf {

constexpr int kWave = 64;
constexpr uint32_t kEndFlag = 0x80000000u;  // top bit of a stream column id = last nnz of its row

template <typename T, int N>
struct alignas(sizeof(T) * N) Vec {
  T v[N];
};

template <typename T>
struct VecOf;  // 16-byte vector of T
template <>
struct VecOf<float> {
  static constexpr int N = 4;
  using type = Vec<float, 4>;
};
template <>
struct VecOf<double> {
  static constexpr int N = 2;
  using type = Vec<double, 2>;
};

__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline int64_t round_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

// 64-lane shuffles for 4- and 8-byte types (all lanes must execute them).
__device__ inline int shfl_i(int v, int src) { return __shfl(v, src, kWave); }
__device__ inline float shfl_t(float v, int src) { return __shfl(v, src, kWave); }
__device__ inline double shfl_t(double v, int src) { return __shfl(v, src, kWave); }
__device__ inline float shfl_xor_t(float v, int m) { return __shfl_xor(v, m, kWave); }
__device__ inline double shfl_xor_t(double v, int m) { return __shfl_xor(v, m, kWave); }

// ---- value of lane L (0..LPR-1) of the caller's LPR-lane group, without touching the LDS pipe:
// DPP quad_perm (LPR 4), DPP row_newbcast (8, 16: a DPP row is 16 lanes), v_readlane (32, 64).
// ds_bpermute costs ~8 LDS-pipe cycles per wave instruction, which bounded the SpMM kernel.
template <int LPR, int L>
__device__ inline int group_bcast_i(int x, int lane) {
  static_assert(L >= 0 && L < LPR, "lane index out of the group");
  // mov_dpp (no "old" operand): every lane is written, so there is no destination to initialise first
  if constexpr (LPR == 4) {
    return __builtin_amdgcn_mov_dpp(x, L | (L << 2) | (L << 4) | (L << 6), 0xf, 0xf, true);
  } else if constexpr (LPR == 8) {
    const int a = __builtin_amdgcn_mov_dpp(x, 0x150 + L, 0xf, 0xf, true);
    const int b = __builtin_amdgcn_mov_dpp(x, 0x150 + 8 + L, 0xf, 0xf, true);
    return (lane & 8) ? b : a;
  } else if constexpr (LPR == 16) {
    return __builtin_amdgcn_mov_dpp(x, 0x150 + L, 0xf, 0xf, true);
  } else if constexpr (LPR == 32) {
    const int a = __builtin_amdgcn_readlane(x, L);
    const int b = __builtin_amdgcn_readlane(x, 32 + L);
    return (lane & 32) ? b : a;
  } else {
    return __builtin_amdgcn_readlane(x, L);
  }
}
template <int LPR, int L>
__device__ inline float group_bcast_t(float x, int lane) {
  return __int_as_float(group_bcast_i<LPR, L>(__float_as_int(x), lane));
}
template <int LPR, int L>
__device__ inline double group_bcast_t(double x, int lane) {
  const int lo = group_bcast_i<LPR, L>(__double2loint(x), lane);
  const int hi = group_bcast_i<LPR, L>(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(integral_constant<int, N-1>{})
template <typename F, int... I>
__device__ inline void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ inline void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define MNMF_HIP_TRY(ctx, expr)                                                            \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) {                                                               \
      (ctx)->fail(MNMF_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__) + " at " + \
                                 __FILE__ + ":" + std::to_string(__LINE__));               \
      return MNMF_EHIP;                                                                    \
    }                                                                                      \
  } while (0)

#define MNMF_TRY(expr)       \
  do {                       \
    int r__ = (expr);        \
    if (r__ != 0) return r__; \
  } while (0)

}  // namespace mnmf
