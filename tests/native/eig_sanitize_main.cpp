// Host-only code of the library under AddressSanitizer + UBSan (tests/test_host_sanitize.py builds and runs this on the
// CPU; the GPU pool has no sanitizer support).  Exercises both eigensolver routes on random, degenerate and tiny inputs.
#include <cstdio>
#include <random>
#define SI_EIG_NS san
#include "../../subspaceinference.jl_amd/csrc/eig.cpp"

using namespace si::san;

static int check(int n, int m, int kind, unsigned seed) {
  std::mt19937_64 r(seed);
  std::normal_distribution<double> nd;
  const int rows = 3 * n + 5;
  std::vector<double> A((size_t)rows * n);
  for (int i = 0; i < rows; ++i) {
    double c = 0;
    for (int j = 0; j < n; ++j) {
      const double v = nd(r);
      c += v;
      A[(size_t)i * n + j] = kind == 0 ? c : v;
    }
    if (kind == 2 && n > 3) A[(size_t)i * n + 1] = A[(size_t)i * n];  // duplicated column
  }
  std::vector<double> g((size_t)n * n, 0.0);
  if (kind != 3)
    for (int i = 0; i < rows; ++i)
      for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) g[a + (size_t)n * b] += A[(size_t)i * n + a] * A[(size_t)i * n + b];
  std::vector<double> full(g), w(n), wt(std::max(m, 1)), V((size_t)n * std::max(m, 1));
  if (sym_eig(n, full.data(), w.data()) != 0) return 1;
  const int rc = sym_eig_top(n, g.data(), m, wt.data(), V.data());
  if (rc == 0)
    for (int k = 0; k < m; ++k)
      if (std::fabs(wt[k] - w[n - 1 - k]) > 1e-9 * (std::fabs(w[n - 1]) + 1e-300)) return 2;
  return 0;
}

int main() {
  int bad = 0;
  const int ns[] = {1, 2, 3, 8, 9, 31, 64, 100};
  for (int n : ns)
    for (int kind = 0; kind < 4; ++kind)
      for (int m : {1, 2, n / 4, n / 3, n}) {
        if (m < 1 || m > n) continue;
        const int rc = check(n, m, kind, 17u * n + kind);
        if (rc != 0) {
          std::printf("FAILED n=%d m=%d kind=%d rc=%d\n", n, m, kind, rc);
          ++bad;
        }
      }
  std::printf(bad ? "EIG_SANITIZE_FAILED\n" : "EIG_SANITIZE_OK\n");
  return bad ? 1 : 0;
}
