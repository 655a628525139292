// H1: dense symmetric eigensolver for the small K x K Gram matrix (host, fp64).
// Part of the replacement of `psvd(A)` (reference src/subspace_construction.jl:63): eigenpairs of A'A give
// V and s^2.  Classic two-stage method: Householder reduction to tridiagonal form with accumulation of the
// orthogonal transform, then the implicit-shift QL iteration on the tridiagonal matrix.  K is at most a few
// thousand, so O(K^3) on one host core is negligible next to the N x K device work.
#include <algorithm>
#include <cmath>
#include <vector>

namespace si {

namespace {

inline double& at(double* a, int n, int i, int j) { return a[i + (size_t)n * j]; }

// Householder tridiagonalisation.  On exit `a` holds the accumulated orthogonal matrix Q (columns), d the
// diagonal and e the sub-diagonal (e[0] = 0) of T = Q' A Q.
void tridiagonalize(int n, double* a, double* d, double* e) {
  for (int j = 0; j < n; ++j) d[j] = at(a, n, n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) {
        d[j] = at(a, n, i - 1, j);
        at(a, n, i, j) = 0.0;
        at(a, n, j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; ++k) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      // apply the similarity transform to the leading block
      for (int j = 0; j < i; ++j) {
        f = d[j];
        at(a, n, j, i) = f;
        g = e[j] + at(a, n, j, j) * f;
        for (int k = j + 1; k <= i - 1; ++k) {
          g += at(a, n, k, j) * d[k];
          e[k] += at(a, n, k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; ++k) at(a, n, k, j) -= (f * e[k] + g * d[k]);
        d[j] = at(a, n, i - 1, j);
        at(a, n, i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  // accumulate transformations
  for (int i = 0; i < n - 1; ++i) {
    at(a, n, n - 1, i) = at(a, n, i, i);
    at(a, n, i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = at(a, n, k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += at(a, n, k, i + 1) * at(a, n, k, j);
        for (int k = 0; k <= i; ++k) at(a, n, k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) at(a, n, k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; ++j) {
    d[j] = at(a, n, n - 1, j);
    at(a, n, n - 1, j) = 0.0;
  }
  at(a, n, n - 1, n - 1) = 1.0;
  e[0] = 0.0;
}

// implicit QL on (d, e), rotating the columns of `a`.  Returns 0 on convergence.
int ql_implicit(int n, double* a, double* d, double* e) {
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::ldexp(1.0, -52);
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m > l) {
      int iter = 0;
      do {
        if (++iter > 200) return 1;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        const double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; ++k) {
            h = at(a, n, k, i + 1);
            at(a, n, k, i + 1) = s * at(a, n, k, i) + c * h;
            at(a, n, k, i) = c * at(a, n, k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
  return 0;
}

}  // namespace

// a: n x n symmetric, column-major; overwritten with eigenvectors (columns); w: eigenvalues ascending.
int sym_eig(int n, double* a, double* w) {
  if (n <= 0) return 0;
  std::vector<double> e(n);
  if (n == 1) {
    w[0] = a[0];
    a[0] = 1.0;
    return 0;
  }
  tridiagonalize(n, a, w, e.data());
  if (ql_implicit(n, a, w, e.data()) != 0) return 1;
  // selection sort of eigenpairs, ascending
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    double p = w[i];
    for (int j = i + 1; j < n; ++j)
      if (w[j] < p) {
        k = j;
        p = w[j];
      }
    if (k != i) {
      w[k] = w[i];
      w[i] = p;
      for (int r = 0; r < n; ++r) std::swap(at(a, n, r, i), at(a, n, r, k));
    }
  }
  return 0;
}

}  // namespace si

extern "C" int si_host_sym_eig(int n, double* a, double* w) { return si::sym_eig(n, a, w); }
