// Run-time choice between the three builds of eig.cpp (baseline x86-64, AVX2+FMA, AVX-512), and the C entry points.
#include <cmath>
#include <utility>

#include "si_internal.h"

namespace si {
namespace base {
int sym_eig(int n, double* a, double* w);
int sym_eig_top(int n, const double* g, int m, double* w_top, double* V);
}  // namespace base
namespace avx2 {
int sym_eig(int n, double* a, double* w);
int sym_eig_top(int n, const double* g, int m, double* w_top, double* V);
}  // namespace avx2
namespace avx512 {
int sym_eig(int n, double* a, double* w);
int sym_eig_top(int n, const double* g, int m, double* w_top, double* V);
}  // namespace avx512

static int isa_level() {  // 2 = AVX-512, 1 = AVX2 + FMA, 0 = baseline
  static const int lvl = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma")) ? 2
                         : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
  return lvl;
}

int sym_eig(int n, double* a, double* w) {
  switch (isa_level()) {
    case 2: return avx512::sym_eig(n, a, w);
    case 1: return avx2::sym_eig(n, a, w);
    default: return base::sym_eig(n, a, w);
  }
}

int sym_eig_top(int n, const double* g, int m, double* w_top, double* V) {
  switch (isa_level()) {
    case 2: return avx512::sym_eig_top(n, g, m, w_top, V);
    case 1: return avx2::sym_eig_top(n, g, m, w_top, V);
    default: return base::sym_eig_top(n, g, m, w_top, V);
  }
}

// Cyclic two-sided Jacobi for a symmetric positive semi-definite matrix with the SCALED stopping criterion
// |a_pq| <= eps * sqrt(a_pp * a_qq) (Demmel & Veselic, "Jacobi's method is more accurate than QR", 1992): eigenvalues
// come out with an error relative to THEMSELVES governed by the condition of D^-1 A D^-1 (D = sqrt(diag A)), not by
// lambda_max -- which is what the second-stage Gram matrix of the ill-conditioned route needs (its columns are graded
// over many decades; a tridiagonalisation would smear eps * lambda_max over the small eigenvalues again).
// a: n x n column-major symmetric, destroyed; w: eigenvalues DESCENDING; v: n x n eigenvectors (columns, same order).
int jacobi_eig_psd(int n, double* a, double* w, double* v) {
  const double eps = 2.220446049250313e-16;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) v[(size_t)j * n + i] = i == j ? 1.0 : 0.0;
  bool converged = false;
  for (int sweep = 0; sweep < 60 && !converged; ++sweep) {
    converged = true;
    for (int p = 0; p < n - 1; ++p) {
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)q * n + p];
        if (apq == 0.0) continue;
        const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
        if (std::fabs(apq) <= eps * std::sqrt(std::fabs(app * aqq))) continue;
        converged = false;
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; ++k) {  // columns p, q of the symmetric matrix (rows follow by symmetry below)
          const double akp = a[(size_t)p * n + k], akq = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * akp - sn * akq;
          a[(size_t)q * n + k] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = a[(size_t)k * n + p], aqk = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * apk - sn * aqk;
          a[(size_t)k * n + q] = sn * apk + c * aqk;
        }
        a[(size_t)p * n + p] = app - t * apq;
        a[(size_t)q * n + q] = aqq + t * apq;
        a[(size_t)q * n + p] = 0.0;
        a[(size_t)p * n + q] = 0.0;
        for (int k = 0; k < n; ++k) {
          const double vkp = v[(size_t)p * n + k], vkq = v[(size_t)q * n + k];
          v[(size_t)p * n + k] = c * vkp - sn * vkq;
          v[(size_t)q * n + k] = sn * vkp + c * vkq;
        }
      }
    }
  }
  // sort descending (selection sort on the diagonal, columns of v follow)
  for (int i = 0; i < n; ++i) w[i] = a[(size_t)i * n + i];
  for (int i = 0; i < n - 1; ++i) {
    int mx = i;
    for (int j = i + 1; j < n; ++j)
      if (w[j] > w[mx]) mx = j;
    if (mx != i) {
      std::swap(w[i], w[mx]);
      for (int k = 0; k < n; ++k) std::swap(v[(size_t)i * n + k], v[(size_t)mx * n + k]);
    }
  }
  return converged ? 0 : 1;
}

}  // namespace si

extern "C" int si_host_jacobi_eig_psd(int n, double* a, double* w, double* v) { return si::jacobi_eig_psd(n, a, w, v); }
extern "C" int si_host_sym_eig(int n, double* a, double* w) { return si::sym_eig(n, a, w); }
// host copy pool sizing (host_copy.cpp), exported for the CPU tests
extern "C" int si_host_cpu_budget(void) { return si::host_cpu_budget(); }
extern "C" double si_host_parse_cpu_max(const char* text) { return si::parse_cpu_max(text); }
extern "C" int si_host_copy_plan(int budget, int nproc, const char* env) { return si::host_copy_plan(budget, nproc, env); }
extern "C" int si_host_sym_eig_top(int n, const double* g, int m, double* w_top, double* V) {
  return si::sym_eig_top(n, g, m, w_top, V);
}
