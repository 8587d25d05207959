// Gather-rate probe: how many bytes per clock per CU the vector memory path delivers when every wave
// gathers 256-byte rows of a table, by load width (dword: 1 row per wave instruction, dwordx2: 2 rows,
// dwordx4: 4 rows) and by table size (L1-, L2-, MALL-resident).  Calibration for DESIGN.md, not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int W>  // W floats per lane; 64/W lanes per row of 64 floats
__global__ __launch_bounds__(256) void probe(const float* __restrict__ tab, const uint32_t* __restrict__ idx,
                                             int n_idx_mask, int iters, float* out) {
  constexpr int LPR = 64 / W, NG = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int g = lane / LPR, c = (lane % LPR) * W;
  float acc[W];
  for (int x = 0; x < W; ++x) acc[x] = 0.f;
  uint32_t p = wave * 977u;
  for (int it = 0; it < iters; it += 8) {
    uint32_t r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = idx[(p + (it + u) * NG + g) & n_idx_mask];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* src = tab + (size_t)r[u] * 64 + c;
      if (W == 4) { float4 v = *(const float4*)src; acc[0] += v.x; acc[1 % W] += v.y; acc[2 % W] += v.z; acc[3 % W] += v.w; }
      else if (W == 2) { float2 v = *(const float2*)src; acc[0] += v.x; acc[1 % W] += v.y; }
      else acc[0] += *src;
    }
  }
  float s = 0; for (int x = 0; x < W; ++x) s += acc[x];
  if (s == 123.456f) out[0] = s;
}

template <int W>
void run(const float* tab, const uint32_t* idx, int mask, float* out, const char* name, size_t tab_rows) {
  const int blocks = 256 * 8, iters = 4096;   // 8 blocks x 4 waves per CU
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  probe<W><<<blocks, 256>>>(tab, idx, mask, iters, out);
  CK(hipEventRecord(a));
  probe<W><<<blocks, 256>>>(tab, idx, mask, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double rows = (double)blocks * 4 * iters * (64 / (64 / W));
  const double bytes = rows * 256;
  printf("%-8s table %8zu rows (%7.1f MB): %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.4GHz  %.2f Grows/s\n", name, tab_rows,
         tab_rows * 256 / 1e6, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9, rows / ms / 1e6);
}

int main() {
  const int NIDX = 1 << 22;
  float* out; CK(hipMalloc(&out, 64));
  for (size_t rows : {64ul, 4096ul, 100000ul, 1000000ul}) {
    float* tab; CK(hipMalloc(&tab, rows * 256)); CK(hipMemset(tab, 0, rows * 256));
    std::vector<uint32_t> h(NIDX);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s % rows); }
    uint32_t* idx; CK(hipMalloc(&idx, NIDX * 4)); CK(hipMemcpy(idx, h.data(), NIDX * 4, hipMemcpyHostToDevice));
    run<1>(tab, idx, NIDX - 1, out, "dword", rows);
    run<2>(tab, idx, NIDX - 1, out, "dwordx2", rows);
    run<4>(tab, idx, NIDX - 1, out, "dwordx4", rows);
    CK(hipFree(tab)); CK(hipFree(idx));
  }
  return 0;
}
