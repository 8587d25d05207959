// Host-side largest eigenvalue of the small symmetric Gram matrices used by the convergence test.
// sigma_max(D) = sqrt(lambda_max(D^T D))  replaces  np.linalg.norm(np.abs(w - w_old), 2)
// (nmf_basic.py:52-53, nmf_projective_basic.py:83) without an SVD of the m x k matrix.
#pragma once
#include <cmath>
#include <vector>

namespace mnmf {

// Cyclic Jacobi on a copy of a (n x n, symmetric, row-major); returns the largest eigenvalue.
inline double sym_lambda_max_jacobi(const double* a_in, int n) {
  if (n <= 0) return 0.0;
  std::vector<double> a(a_in, a_in + (size_t)n * n);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += a[(size_t)i * n + i] * a[(size_t)i * n + i];
      for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
    }
    if (off <= 1e-30 * (diag + off) || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p) {
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int r = 0; r < n; ++r) {  // columns p, q
          const double arp = a[(size_t)r * n + p], arq = a[(size_t)r * n + q];
          a[(size_t)r * n + p] = c * arp - s * arq;
          a[(size_t)r * n + q] = s * arp + c * arq;
        }
        for (int r = 0; r < n; ++r) {  // rows p, q
          const double apr = a[(size_t)p * n + r], aqr = a[(size_t)q * n + r];
          a[(size_t)p * n + r] = c * apr - s * aqr;
          a[(size_t)q * n + r] = s * apr + c * aqr;
        }
      }
    }
  }
  double best = a[0];
  for (int i = 1; i < n; ++i) best = std::max(best, a[(size_t)i * n + i]);
  return best;
}

// lambda_max of an entrywise non-negative symmetric PSD matrix (Gram of |delta|) by power
// iteration with two-sided bounds: Rayleigh quotient (lower) and Collatz-Wielandt max_i (Gx)_i/x_i
// (upper, valid for G >= 0 and x > 0).  Falls back to Jacobi if the bounds do not close.
inline double nonneg_lambda_max(const double* g, int n) {
  std::vector<int> idx;
  for (int i = 0; i < n; ++i)
    if (g[(size_t)i * n + i] > 0.0) idx.push_back(i);
  const int r = (int)idx.size();
  if (r == 0) return 0.0;
  std::vector<double> x(r, 1.0), y(r);
  double lo = 0.0, hi = INFINITY;
  for (int it = 0; it < 3000; ++it) {
    double xx = 0.0, xy = 0.0, up = 0.0;
    for (int i = 0; i < r; ++i) {
      double s = 0.0;
      const double* row = g + (size_t)idx[i] * n;
      for (int j = 0; j < r; ++j) s += row[idx[j]] * x[j];
      y[i] = s;
      xx += x[i] * x[i];
      xy += x[i] * s;
      up = std::max(up, s / x[i]);
    }
    lo = std::max(lo, xy / xx);
    hi = std::min(hi, up);
    if (hi - lo <= 1e-13 * hi) return 0.5 * (lo + hi);
    double ymax = 0.0;
    for (int i = 0; i < r; ++i) ymax = std::max(ymax, y[i]);
    if (!(ymax > 0.0) || !std::isfinite(ymax)) break;
    bool positive = true;
    for (int i = 0; i < r; ++i) {
      x[i] = y[i] / ymax;
      if (!(x[i] > 0.0)) positive = false;
    }
    if (!positive) break;  // underflow of a component: bounds no longer valid
  }
  if (std::isfinite(hi) && hi - lo <= 1e-9 * hi) return 0.5 * (lo + hi);
  return sym_lambda_max_jacobi(g, n);
}

}  // namespace mnmf
