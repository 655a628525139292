// Run-time choice between the two builds of eig.cpp (baseline x86-64 and AVX2+FMA), and the C entry points.
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

static bool have_avx2() {
  static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  return ok;
}

int sym_eig(int n, double* a, double* w) { return have_avx2() ? avx2::sym_eig(n, a, w) : base::sym_eig(n, a, w); }

int sym_eig_top(int n, const double* g, int m, double* w_top, double* V) {
  return have_avx2() ? avx2::sym_eig_top(n, g, m, w_top, V) : base::sym_eig_top(n, g, m, w_top, V);
}

}  // namespace si

extern "C" int si_host_sym_eig(int n, double* a, double* w) { return si::sym_eig(n, a, w); }
extern "C" int si_host_sym_eig_top(int n, const double* g, int m, double* w_top, double* V) {
  return si::sym_eig_top(n, g, m, w_top, V);
}
