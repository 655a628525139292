// H1: dense symmetric eigensolver for the small K x K Gram matrix (host, fp64).
// Part of the replacement of `psvd(A)` (reference src/subspace_construction.jl:63): eigenpairs of A'A give
// V and s^2.  Classic two-stage method: Householder reduction to tridiagonal form with accumulation of the
// orthogonal transform, then the implicit-shift QL iteration on the tridiagonal matrix.  K is at most a few
// thousand, so O(K^3) on one host core is negligible next to the N x K device work.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

// This file is compiled three times (build.py): as SI_EIG_NS=base with the default x86-64 flags, as SI_EIG_NS=avx2 with
// -mavx2 -mfma and as SI_EIG_NS=avx512 with -mavx512f -mfma (the inner loops are contiguous column sweeps and the
// lane-parallel bisection: wider vectors cut the time); eig_dispatch.cpp picks one at run time from the CPU's feature
// bits.
#ifndef SI_EIG_NS
#define SI_EIG_NS base
#endif

namespace si {
namespace SI_EIG_NS {

namespace {

inline double& at(double* a, int n, int i, int j) { return a[i + (size_t)n * j]; }

// Dot products with 16 independent partial sums (two vectors of 8 lanes): a single running sum is one chain of dependent
// additions, 4 cycles each whatever the vector width -- the Householder reduction of a 100 x 100 matrix spent 0.3 ms in
// exactly that chain.  The order of the additions is fixed (lane j takes elements j, j + 16, ...; lanes summed pairwise in a
// fixed tree), so results are reproducible from run to run.
typedef double v8d __attribute__((vector_size(64)));      // explicit 8-lane vectors: hipcc's clang compiles this file, and
inline v8d ld8(const double* p) {                           // left loops over an array of partial sums scalar
  v8d v;
  __builtin_memcpy(&v, p, sizeof(v8d));
  return v;
}
inline void st8(double* p, v8d v) { __builtin_memcpy(p, &v, sizeof(v8d)); }
inline double lanes_sum(v8d a, v8d b) {
  const v8d t = a + b;
  return ((t[0] + t[4]) + (t[1] + t[5])) + ((t[2] + t[6]) + (t[3] + t[7]));
}
inline double dot_lanes(const double* x, const double* y, int n) {   // (x and y may be the same vector)
  v8d a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
  int i = 0;
  for (; i + 16 <= n; i += 16) {
    a0 += ld8(x + i) * ld8(y + i);
    a1 += ld8(x + i + 8) * ld8(y + i + 8);
  }
  if (i + 8 <= n) {
    a0 += ld8(x + i) * ld8(y + i);
    i += 8;
  }
  double tail = 0.0;
  for (; i < n; ++i) tail += x[i] * y[i];
  return lanes_sum(a0, a1) + tail;
}
// acc = sum_r col[r] * x[r]  and  p[r] += col[r] * vc  in one sweep over col (one column of the symmetric rank-2 step)
inline double dot_axpy_lanes(const double* __restrict col, const double* __restrict x, double* __restrict p, double vc, int n) {
  v8d a0 = {0, 0, 0, 0, 0, 0, 0, 0};
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    const v8d cv = ld8(col + i);
    a0 += cv * ld8(x + i);
    st8(p + i, ld8(p + i) + cv * vc);
  }
  double tail = 0.0;
  for (; i < n; ++i) {
    tail += col[i] * x[i];
    p[i] += col[i] * vc;
  }
  const v8d zero = {0, 0, 0, 0, 0, 0, 0, 0};
  return lanes_sum(a0, zero) + tail;
}
// Whole-vector forms for the Householder reduction (n8 = a multiple of 8, no scalar tails: at K ~ 100 the column loops
// average 50 elements and their heads and tails cost more than their bodies).  The caller guarantees that the operands may be
// read (and p / col written) up to n8 and that x (resp. vp, wp) holds ZEROS past the logical length, so the extra lanes add
// exact zeros to the sums and leave the extra matrix entries -- the unused upper triangle, or padding -- unchanged.
inline double dot_axpy8(const double* __restrict col, const double* __restrict x, double* __restrict p, double vc, int n8) {
  v8d acc = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n8; i += 8) {
    const v8d cv = ld8(col + i);
    acc += cv * ld8(x + i);
    st8(p + i, ld8(p + i) + cv * vc);
  }
  return ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
}
inline void rank2_update8(double* __restrict col, const double* __restrict vp, const double* __restrict wp, double vc, double wc,
                          int n8) {
  for (int i = 0; i < n8; i += 8) st8(col + i, ld8(col + i) - (ld8(vp + i) * wc + ld8(wp + i) * vc));
}
// y -= a * x (the two never overlap: said so, or the compiler keeps the loop scalar)
inline void axpy_neg(double* __restrict y, const double* __restrict x, double a, int n) {
  for (int i = 0; i < n; ++i) y[i] -= a * x[i];
}

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

// ---- top-m eigenpairs without forming all n eigenvectors -------------------------------------------------------
// si_construct_finish needs the M largest eigenpairs of the K x K Gram matrix (M = 20 of K = 100 at cfg2).  The
// full solver above spends most of its time accumulating Q and rotating all K columns in the QL sweeps (~7 K^3
// flops); this route does the Householder reduction with the reflectors kept in factored form (4/3 K^3), takes the
// eigenvalues by bisection on the Sturm count with all m targets in flight (vectorised), gets the m wanted
// eigenvectors of T by inverse iteration (O(K) each,
// re-orthogonalised inside clusters of close eigenvalues) and applies the reflectors to those m vectors only
// (2 K^2 m).  The result is VERIFIED (residual and orthogonality) by the caller-visible wrapper, which falls back to
// the full solver if the check fails, so degenerate spectra cost time, never accuracy.

// lower-triangle Householder reduction, reflectors in factored form (as LAPACK dsytd2, uplo = 'L'):
// on exit d, e (e[i] couples i and i+1) describe T; column i of `a` below the sub-diagonal holds v_i[2:] (v_i[1] = 1)
// and tau[i] its scale;  Q = H_0 H_1 ... H_{n-3},  H_i = I - tau_i v_i v_i'.
void tridiagonalize_factored(int n, double* a, double* d, double* e, double* tau, double* work) {
  // One sweep over the trailing block per step instead of two: the symmetric rank-2 update of step i-1 (A -= v w' + w v') is
  // applied to a column in the same pass that accumulates step i's product A v' from it -- half the passes over the matrix and
  // half the short column loops (at K = 100 .. 700 those loops, not the arithmetic, are what the reduction costs).
  const int np = n + 16;          // every work vector has 16 spare elements: the 8-lane loops run past the logical length
  double* pvec = work;            // [np]  p = tau A v, then w
  double* vprev = work + np;      // [np]  reflector of the previous step (v[0] = 1 explicit), indexed from its first row
  double* wprev = work + 2 * np;  // [np]  its w
  double* vcur = work + 3 * np;   // [np]  reflector of this step, v[0] = 1 explicit
  for (int k = 0; k < 4 * np; ++k) work[k] = 0.0;
  bool pending = false;           // (vprev, wprev) of length n - i still to be applied to A[i.., i..]
  for (int i = 0; i < n - 1; ++i) {
    const int s = n - i - 1;            // length of the column below the diagonal
    double* ci = &at(a, n, i, i);       // column i from its diagonal: ci[0] = A(i,i), ci[1..s] = x
    if (pending) {                      // bring column i up to date: rows i .. n-1 are vprev / wprev [0 .. s]
      const double v0 = vprev[0], w0 = wprev[0];
      for (int r = 0; r <= s; ++r) ci[r] -= vprev[r] * w0 + wprev[r] * v0;
    }
    d[i] = ci[0];
    double* x = ci + 1;                 // x[0..s)
    const double xnorm2 = dot_lanes(x + 1, x + 1, s - 1);
    const double alpha = x[0];
    double t = 0.0, beta = alpha;
    if (xnorm2 > 0.0) {
      beta = -std::copysign(std::sqrt(alpha * alpha + xnorm2), alpha);
      t = (beta - alpha) / beta;
      const double inv = 1.0 / (alpha - beta);
      for (int k = 1; k < s; ++k) x[k] *= inv;
    }
    e[i] = beta;
    tau[i] = t;
    x[0] = beta;                        // T's sub-diagonal in place of v[0] (v[0] = 1 is implicit in the stored reflector)
    vcur[0] = 1.0;
    for (int k = 1; k < s; ++k) vcur[k] = x[k];
    for (int k = s; k < s + 8; ++k) vcur[k] = 0.0;   // zeros past the end (the previous, longer reflector lived here)
    const bool symv = t != 0.0;
    if (symv)
      for (int r = 0; r < s + 8; ++r) pvec[r] = 0.0;
    if (pending || symv) {
      // trailing block A[i+1.., i+1..], lower triangle by columns: update with (vprev, wprev)[1 + .], product with vcur
      const double* vp = vprev + 1;
      const double* wp = wprev + 1;
      for (int c = 0; c < s; ++c) {
        double* col = &at(a, n, i + 1 + c, i + 1 + c);  // col[0] = diagonal, col[r - c] for r > c
        const int len = s - c;
        if (pending) rank2_update8(col, vp + c, wp + c, vp[c], wp[c], (len + 7) & ~7);
        if (symv) {
          const double vc = vcur[c];
          pvec[c] += col[0] * vc + dot_axpy8(col + 1, vcur + c + 1, pvec + c + 1, vc, (len - 1 + 7) & ~7);
        }
      }
    }
    pending = symv;
    if (symv) {
      for (int r = 0; r < s; ++r) pvec[r] *= t;
      const double pv = dot_lanes(pvec, vcur, s);
      const double half = 0.5 * t * pv;
      for (int r = 0; r < s; ++r) {
        wprev[r] = pvec[r] - half * vcur[r];   // w = p - (tau/2)(p'v) v
        vprev[r] = vcur[r];
      }
      for (int r = s; r < s + 8; ++r) wprev[r] = vprev[r] = 0.0;   // zeros past the end for the 8-lane update of the next step
    }
  }
  if (pending) at(a, n, n - 1, n - 1) -= 2.0 * vprev[0] * wprev[0];
  d[n - 1] = at(a, n, n - 1, n - 1);
  if (n >= 2) tau[n - 2] = 0.0;  // the last "reflector" acts on a single element: H = I
}

// eigenvalues only: implicit QL on copies of (d, e[0..n-1) sub-diagonal); ascending on exit.  Returns 0 on convergence.
// The m largest eigenvalues of the symmetric tridiagonal T (d, e: e[i] couples i and i+1) by bisection on the Sturm
// count, ALL m targets advanced together: the inner loop over the targets has no dependency between its lanes (one
// division each) and vectorises, where a vector-free QL sweep is one serial chain of sqrt and divisions (it took 40 %
// of the whole top-m route).  w[k] = k-th largest, to the precision of the count (a few ulp of ||T||).
void bisect_top(int n, const double* d, const double* e, int m, double* w, double* work /* 5 * mp doubles */) {
  const int mp = (m + 7) / 8 * 8;
  double *lo = work, *hi = lo + mp, *sig = hi + mp, *q = sig + mp, *cnt = q + mp;
  double gl = d[0], gu = d[0], tnorm = 0.0;
  for (int i = 0; i < n; ++i) {
    const double r = (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(e[i]) : 0.0);
    gl = std::min(gl, d[i] - r);
    gu = std::max(gu, d[i] + r);
    tnorm = std::max(tnorm, std::fabs(d[i]) + r);
  }
  const double eps = std::ldexp(1.0, -52);
  const double pivmin = std::max(eps * eps * tnorm * tnorm, 1e-300);  // floor for |q| (a square: q carries e^2 / q)
  const double pad = 2.0 * eps * tnorm * n + 1e-300;
  for (int k = 0; k < mp; ++k) {
    lo[k] = gl - pad;
    hi[k] = gu + pad;
  }
  std::vector<double> e2(n, 0.0);
  for (int i = 0; i + 1 < n; ++i) e2[i] = e[i] * e[i];
  for (int round = 0; round < 58; ++round) {  // the interval starts ~2 ||T|| wide: 58 halvings reach 1e-17 ||T||
    for (int k = 0; k < mp; ++k) {
      sig[k] = 0.5 * (lo[k] + hi[k]);
      double t = d[0] - sig[k];
      t = std::fabs(t) < pivmin ? -pivmin : t;
      q[k] = t;
      cnt[k] = t < 0.0 ? 1.0 : 0.0;
    }
    for (int i = 1; i < n; ++i) {
      const double di = d[i], ei2 = e2[i - 1];
      for (int k = 0; k < mp; ++k) {  // independent lanes: vectorised
        double t = di - sig[k] - ei2 / q[k];
        t = std::fabs(t) < pivmin ? -pivmin : t;
        q[k] = t;
        cnt[k] += t < 0.0 ? 1.0 : 0.0;
      }
    }
    // target k is the (n-1-k)-th eigenvalue in ascending order: the smallest sigma with count(sigma) >= n - k
    for (int k = 0; k < mp; ++k) {
      const double need = (double)(n - (k < m ? k : m - 1));
      if (cnt[k] >= need)
        hi[k] = sig[k];
      else
        lo[k] = sig[k];
    }
  }
  for (int k = 0; k < m; ++k) w[k] = 0.5 * (lo[k] + hi[k]);
}

// LU factorisation with partial pivoting of the tridiagonal T - lambda I (d, e: e[i] couples i and i+1), then solves
// with it.  Tiny pivots are replaced by `pivmin` (inverse iteration WANTS the near-singularity).  U has two
// super-diagonals (u1, u2) because of the row swaps; l holds the multipliers, swapped the row exchanges.
void tridiag_shift_factor(int n, const double* d, const double* e, double lambda, double pivmin, double* u0, double* u1,
                          double* u2, double* l, int* swapped) {
  for (int i = 0; i < n; ++i) {
    u0[i] = d[i] - lambda;
    u1[i] = i + 1 < n ? e[i] : 0.0;
    u2[i] = 0.0;
  }
  for (int i = 0; i + 1 < n; ++i) {
    const double sub = e[i];  // element (i+1, i)
    if (std::fabs(u0[i]) >= std::fabs(sub)) {
      swapped[i] = 0;
      double piv = u0[i];
      if (std::fabs(piv) < pivmin) piv = u0[i] = std::copysign(pivmin, piv == 0.0 ? 1.0 : piv);
      const double mlt = sub / piv;
      l[i] = mlt;
      u0[i + 1] -= mlt * u1[i];
    } else {
      swapped[i] = 1;  // rows i and i+1 change places: the pivot row is (sub, d[i+1]-lambda, e[i+1])
      const double mlt = u0[i] / sub;
      l[i] = mlt;
      const double r0 = u0[i + 1], r1 = u1[i + 1];
      const double old_u1 = u1[i];
      u0[i] = sub;
      u1[i] = r0;
      u2[i] = r1;
      u0[i + 1] = old_u1 - mlt * r0;
      u1[i + 1] = -mlt * r1;
    }
  }
  if (std::fabs(u0[n - 1]) < pivmin) u0[n - 1] = std::copysign(pivmin, u0[n - 1] == 0.0 ? 1.0 : u0[n - 1]);
  for (int i = 0; i < n; ++i) u0[i] = 1.0 / u0[i];  // the solves multiply
}

void tridiag_shift_solve(int n, const double* u0inv, const double* u1, const double* u2, const double* l,
                         const int* swapped, double* x) {
  for (int i = 0; i + 1 < n; ++i) {  // forward substitution with the same row operations
    if (swapped[i]) {
      const double t = x[i];
      x[i] = x[i + 1];
      x[i + 1] = t - l[i] * x[i];
    } else {
      x[i + 1] -= l[i] * x[i];
    }
  }
  for (int i = n - 1; i >= 0; --i) {  // back substitution
    double t = x[i];
    if (i + 1 < n) t -= u1[i] * x[i + 1];
    if (i + 2 < n) t -= u2[i] * x[i + 2];
    x[i] = t * u0inv[i];
  }
}

// top-m eigenpairs of the symmetric n x n matrix `a` (column-major, destroyed): w_top[0..m) descending, V n x m.
// Returns 0 = done (unverified: see sym_eig_top), 1 = inverse iteration broke down.
int sym_eig_top_unverified(int n, double* a, int m, double* w_top, double* V) {
  std::vector<double> d(n), e(n), tau(n), work(6 * ((size_t)n + 16));
  std::vector<int> sw(n);
  tridiagonalize_factored(n, a, d.data(), e.data(), tau.data(), work.data());
  std::vector<double> wtop_t((size_t)m), bwork(5 * ((size_t)m + 8));
  bisect_top(n, d.data(), e.data(), m, wtop_t.data(), bwork.data());
  double tnorm = 0.0;  // 1-norm of T
  for (int i = 0; i < n; ++i)
    tnorm = std::max(tnorm, std::fabs(d[i]) + (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(e[i]) : 0.0));
  const double eps = std::ldexp(1.0, -52);
  const double pivmin = std::max(eps * tnorm, 1e-300);
  const double ortol = 1e-3 * tnorm, septol = 10.0 * eps * tnorm;
  double* u0 = work.data();
  double *u1 = u0 + n, *u2 = u1 + n, *l = u2 + n, *x = l + n;
  uint64_t rng = 0x9E3779B97F4A7C15ull;
  int cluster0 = 0;       // first vector of the current cluster
  double lam_prev = 0.0;  // (perturbed) eigenvalue used for the previous vector
  for (int k = 0; k < m; ++k) {
    double lam = wtop_t[(size_t)k];
    w_top[k] = lam;
    if (k > 0) {
      if (lam_prev - lam > ortol) cluster0 = k;                 // well separated: new cluster
      if (lam_prev - lam < septol) lam = lam_prev - septol;      // coincident: separate the shifts
    }
    lam_prev = lam;
    for (int i = 0; i < n; ++i) {  // deterministic start vector
      rng = rng * 6364136223846793005ull + 1442695040888963407ull;
      x[i] = ((double)(rng >> 11) / 9007199254740992.0) - 0.5;
    }
    double* vk = V + (size_t)k * n;
    tridiag_shift_factor(n, d.data(), e.data(), lam, pivmin, u0, u1, u2, l, sw.data());
    for (int it = 0; it < 3; ++it) {
      tridiag_shift_solve(n, u0, u1, u2, l, sw.data(), x);
      for (int c = cluster0; c < k; ++c) {  // modified Gram-Schmidt inside the cluster
        const double* vc = V + (size_t)c * n;
        const double dot = dot_lanes(vc, x, n);
        axpy_neg(x, vc, dot, n);
      }
      const double nrm = std::sqrt(dot_lanes(x, x, n));
      if (!(nrm > 0.0) || !std::isfinite(nrm)) return 1;
      for (int i = 0; i < n; ++i) x[i] /= nrm;
    }
    for (int i = 0; i < n; ++i) vk[i] = x[i];   // eigenvector of T (kept for the cluster orthogonalisation)
  }
  // back-transform  V <- Q V,  Q = H_0 ... H_{n-3}:  apply H_i for i = n-3 .. 0 to rows i+1 .. n-1
  for (int i = n - 3; i >= 0; --i) {
    const double t = tau[i];
    if (t == 0.0) continue;
    const double* v = &at(a, n, i + 1, i);  // v[0] = 1 implicit, v[1..] stored
    const int s = n - i - 1;
    for (int c = 0; c < m; ++c) {
      double* z = V + (size_t)c * n + i + 1;
      const double dot = (z[0] + dot_lanes(v + 1, z + 1, s - 1)) * t;
      z[0] -= dot;
      axpy_neg(z + 1, v + 1, dot, s - 1);
    }
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

// Top-m eigenpairs of the symmetric n x n matrix g (column-major, left intact): w_top descending, V n x m, lam_min_out
// unused (<0) ... verified: every returned pair satisfies ||g v - w v|| <= 1e-12 ||g||_F and the vectors are orthonormal
// to 1e-10; returns 0 ok, 1 = failed (the caller uses the full solver).
int sym_eig_top(int n, const double* g, int m, double* w_top, double* V) {
  if (n <= 0 || m <= 0 || m > n) return 1;
  if (n < 8 || 3 * m > n) return 1;  // no advantage over the full solver
  std::vector<double> a((size_t)n * n + 16, 0.0);   // + padding: the reduction's 8-lane loops run past a column's end
  std::copy(g, g + (size_t)n * n, a.begin());
  if (sym_eig_top_unverified(n, a.data(), m, w_top, V) != 0) return 1;
  double fro = 0.0;
  for (int j = 0; j < n; ++j) fro += dot_lanes(g + (size_t)j * n, g + (size_t)j * n, n);
  fro = std::sqrt(fro);
  std::vector<double> r(n);
  for (int k = 0; k < m; ++k) {
    const double* v = V + (size_t)k * n;
    for (int i = 0; i < n; ++i) r[i] = dot_lanes(g + (size_t)i * n, v, n) - w_top[k] * v[i];   // g is symmetric: row i = column i
    const double res = dot_lanes(r.data(), r.data(), n);
    if (!(std::sqrt(res) <= 1e-12 * fro)) return 1;
    for (int c = 0; c <= k; ++c) {
      const double* vc = V + (size_t)c * n;
      const double dot = dot_lanes(vc, v, n);
      if (!(std::fabs(dot - (c == k ? 1.0 : 0.0)) <= 1e-10)) return 1;
    }
  }
  return 0;
}

}  // namespace SI_EIG_NS
}  // namespace si
