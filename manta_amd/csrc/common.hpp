// Shared helpers for the mnmf engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

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
