// Host-side helpers of libmnmf.so under AddressSanitizer ("compile-time -fsanitize=address host build
// for the C-ABI glue").  The segment builders (sparse_prep.hpp: build_segments_host, build_tile_segments_host) and the
// k x k eigen-solvers (eig.hpp) are pointer- and index-heavy host code; this driver runs them on randomised inputs of
// every mode the engine uses and checks their invariants, with the host code instrumented
// (hipcc -Xarch_host -fsanitize=address; no GPU is touched).  tests/test_host_asan.py builds and runs it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../manta_amd/csrc/eig.hpp"
#include "../../manta_amd/csrc/sparse_prep.hpp"

using namespace mnmf;

static int failures = 0;
#define CHECK(c)                                                         \
  do {                                                                   \
    if (!(c)) {                                                          \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      ++failures;                                                        \
    }                                                                    \
  } while (0)

// every nnz belongs to exactly one segment; pieces end where the list says; carry slots are all used once
static void check_segments(const HostSegments& s, const HostPieces& pc, int64_t rows, int64_t sub) {
  const int64_t nnz = pc.start.back();
  std::vector<int> covered((size_t)nnz, 0);
  int64_t total = 0;
  for (int64_t i = 0; i < s.n_seg(); ++i) {
    CHECK(s.seg_len[i] >= 0);
    for (int64_t p = s.seg_start[i]; p < s.seg_start[i] + s.seg_len[i]; ++p) {
      CHECK(p >= 0 && p < nnz);
      if (p >= 0 && p < nnz) ++covered[(size_t)p];
    }
    total += s.seg_len[i];
    if (s.seg_len[i] > 0) CHECK(s.seg_info[i] >= 0 && s.seg_info[i] < (int64_t)s.dest.size());
  }
  CHECK(total == nnz);
  for (int64_t p = 0; p < nnz; ++p) CHECK(covered[(size_t)p] == 1);
  CHECK(s.dest.size() == s.piece_end.size());
  std::vector<int> slot_used((size_t)s.n_carry, 0);
  int64_t prev = -1;
  std::vector<int64_t> ends(s.piece_end);
  std::sort(ends.begin(), ends.end());
  for (int64_t e : ends) {
    CHECK(e > prev && e < nnz && e - prev <= sub);
    prev = e;
  }
  if (!ends.empty()) CHECK(ends.back() == nnz - 1);
  for (size_t i = 0; i < s.dest.size(); ++i) {
    const int32_t d = s.dest[i];
    if (d >= 0) CHECK(d < rows);
    else {
      CHECK(~d < s.n_carry);
      if (~d < s.n_carry) ++slot_used[(size_t)~d];
    }
  }
  for (int64_t c = 0; c < s.n_carry; ++c) CHECK(slot_used[(size_t)c] == 1);
  int64_t carried = 0;
  for (size_t i = 0; i < s.fix_row.size(); ++i) {
    CHECK(s.fix_row[i] >= 0 && s.fix_row[i] < rows && s.fix_count[i] >= 2);
    CHECK(s.fix_first[i] >= 0 && s.fix_first[i] + s.fix_count[i] <= s.n_carry);
    carried += s.fix_count[i];
  }
  CHECK(carried == s.n_carry);
}

int main() {
  std::mt19937_64 rng(12345);
  auto uni = [&](int64_t lo, int64_t hi) { return (int64_t)(lo + rng() % (uint64_t)(hi - lo + 1)); };
  // ---- plain CSR rows
  for (int rep = 0; rep < 20; ++rep) {
    const int64_t rows = uni(1, 300);
    std::vector<int64_t> rp((size_t)rows + 1, 0);
    for (int64_t r = 0; r < rows; ++r) rp[(size_t)r + 1] = rp[(size_t)r] + (uni(0, 9) == 0 ? uni(500, 2100) : uni(0, 40));
    HostPieces pc;
    pieces_from_rowptr(rp.data(), rows, pc);
    for (int64_t q : {1, 16, 64, 512}) {
      for (int exact = 0; exact < 2; ++exact) {
        HostSegments s;
        build_segments_host(pc, rows, q, kSubPiece, s, 0, nullptr, 0, exact != 0);
        check_segments(s, pc, rows, kSubPiece);
        if (exact)
          for (int64_t i = 0; i + 1 < s.n_seg(); ++i) CHECK(s.seg_len[i] == q);
      }
    }
  }
  // ---- blocked pieces (X^T), XCD-affine order, exact or not; tile pieces
  for (int rep = 0; rep < 20; ++rep) {
    const int64_t rows = uni(2, 120), n_blocks = uni(1, 30);
    HostPieces pc;
    int64_t pos = 0;
    for (int64_t b = 0; b < n_blocks; ++b)
      for (int64_t r = 0; r < rows / 2; ++r)
        if (uni(0, 3) != 0) {
          pc.start.push_back(pos);
          pc.row.push_back((int32_t)r);
          pc.block.push_back((int32_t)b);
          pos += uni(1, 1300);
        }
    for (int64_t r = rows / 2; r < rows; ++r)
      if (uni(0, 4) != 0) {
        pc.start.push_back(pos);
        pc.row.push_back((int32_t)r);
        pc.block.push_back(-1);
        pos += uni(1, 60);
      }
    pc.start.push_back(pos);
    if (pc.row.empty()) continue;
    for (int group : {0, 4, 16, 32})
      for (int64_t q : {32, 128})
        for (int exact = 0; exact < 2; ++exact) {
          HostSegments s;
          build_segments_host(pc, rows, q, kSubPiece, s, group, nullptr, 0, exact != 0);
          check_segments(s, pc, rows, kSubPiece);
          if (group) CHECK(s.n_seg() % (8 * group) == 0);
        }
    // the same pieces as a tile stream: blocks ascending, no loose pieces
    HostPieces tp;
    for (size_t i = 0; i < pc.row.size(); ++i)
      if (pc.block[i] >= 0) {
        tp.start.push_back(pc.start[i] - pc.start[0]);
        tp.row.push_back(pc.row[i]);
        tp.block.push_back(pc.block[i]);
      }
    if (tp.row.empty()) continue;
    {  // re-base the starts so the pieces are contiguous
      int64_t p2 = 0;
      std::vector<int64_t> st;
      for (size_t i = 0; i < tp.row.size(); ++i) {
        st.push_back(p2);
        size_t j = 0;
        while (pc.start[j] - pc.start[0] != tp.start[i]) ++j;
        p2 += pc.start[j + 1] - pc.start[j];
      }
      st.push_back(p2);
      tp.start = st;
    }
    for (int64_t q : {17, 96}) {
      HostSegments s;
      std::vector<int32_t> tf;
      build_segments_host(tp, rows, q, kSubPiece, s, 0, &tf, n_blocks);
      check_segments(s, tp, rows, kSubPiece);
      CHECK((int64_t)tf.size() == n_blocks + 1 && tf.back() == s.n_seg());
    }
    for (int groups : {1, 4, 64}) {
      HostSegments s;
      std::vector<int32_t> tf;
      build_tile_segments_host(tp, rows, n_blocks, groups, kSubPiece, s, tf);
      check_segments(s, tp, rows, kSubPiece);
      CHECK((int64_t)tf.size() == n_blocks + 1 && tf.back() == s.n_seg());
    }
  }
  // ---- eigen-solvers on Gram matrices of non-negative data: both agree, and with a power iteration
  for (int n : {1, 2, 5, 16, 50, 64}) {
    std::vector<double> a((size_t)n * 7), g((size_t)n * n, 0.0);
    for (double& x : a) x = (double)(rng() % 1000) / 1000.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        for (int t = 0; t < 7; ++t) g[(size_t)i * n + j] += a[(size_t)i * 7 + t] * a[(size_t)j * 7 + t];
    const double l1 = sym_lambda_max_jacobi(g.data(), n), l2 = nonneg_lambda_max(g.data(), n);
    CHECK(std::isfinite(l1) && std::isfinite(l2));
    CHECK(std::fabs(l1 - l2) <= 1e-9 * std::max(1.0, std::fabs(l1)));
  }
  if (failures) {
    std::fprintf(stderr, "%d checks failed\n", failures);
    return 1;
  }
  std::printf("OK host helpers under AddressSanitizer\n");
  return 0;
}
