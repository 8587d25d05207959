// Probe for the "synchronised sweep" idea: 256 persistent workgroups gather 256-byte rows of a
// 256 MB table (W at C2).  Every workgroup's row list is sorted into document windows and the workgroups walk the
// windows in step (a barrier per window inside the workgroup, nothing between workgroups), so the workgroups of one
// XCD touch the same window of the table at about the same time and a row fetched by one is an L2 hit for the others.
// Compared with the same number of gathers in random order.   hipcc -O3 --offload-arch=gfx950 sweep_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int L>
__global__ __launch_bounds__(1024) void probe(const float* __restrict__ tab, const unsigned* __restrict__ idx, int n_win, float* out) {
  using U4 = __attribute__((ext_vector_type(4))) unsigned int;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = wv * 4 + lane / 16, li = lane % 16;      // 64 groups of 16 lanes, a lane reads 16 B of a 256-B row
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tab), 0, 0xFFFFFFF0u, 0x00020000);
  float acc = 0.f;
  const unsigned* p = idx + ((size_t)blockIdx.x * n_win * 64 + g) * L;
  for (int w = 0; w < n_win; ++w) {
    unsigned id[L];
#pragma unroll
    for (int x = 0; x < L; ++x) id[x] = p[x];
    U4 v[L];
#pragma unroll
    for (int x = 0; x < L; ++x) v[x] = __builtin_amdgcn_raw_buffer_load_b128(r, id[x] * 256u + li * 16u, 0, 0);
#pragma unroll
    for (int x = 0; x < L; ++x) acc += __uint_as_float(v[x][0]) + __uint_as_float(v[x][3]);
    p += 64 * L;
    __syncthreads();
  }
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int rows = 1000000, L = 12, n_cu = 256;
  const int n_win = argc > 1 ? atoi(argv[1]) : 122;
  float* tab; unsigned* idx; float* out;
  CK(hipMalloc(&tab, (size_t)rows * 256));
  CK(hipMemset(tab, 0, (size_t)rows * 256));
  const size_t n = (size_t)n_cu * n_win * 64 * L;
  CK(hipMalloc(&idx, n * 4));
  CK(hipMalloc(&out, (size_t)n_cu * 1024 * 4));
  std::vector<unsigned> h(n);
  std::mt19937 rng(1);
  for (int mode = 0; mode < 3; ++mode) {   // 0: windows in step, 1: random rows, 2: windows, but every workgroup starts at another window
    for (int c = 0; c < n_cu; ++c)
      for (int w = 0; w < n_win; ++w) {
        const int ww = mode == 2 ? (w + c * 7) % n_win : w;
        const unsigned lo = (unsigned)((long long)rows * ww / n_win), hi = (unsigned)((long long)rows * (ww + 1) / n_win);
        for (int i = 0; i < 64 * L; ++i)
          h[((size_t)c * n_win + w) * 64 * L + i] = mode == 1 ? rng() % rows : lo + rng() % (hi - lo);
      }
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) probe<12><<<n_cu, 1024>>>(tab, idx, n_win, out);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) probe<12><<<n_cu, 1024>>>(tab, idx, n_win, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    printf("mode %d (%s): %zu gathers of 256 B, %d windows: %.3f ms  %.1f G rows/s  %.2f TB/s\n", mode,
           mode == 0 ? "windows in step" : mode == 1 ? "random" : "windows, staggered start", n, n_win, ms, n / ms / 1e6, n * 256.0 / ms / 1e9);
  }
  return 0;
}
