// Run-time choice between the three builds of eig.cpp (baseline x86-64, AVX2+FMA, AVX-512), and the C entry points.
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

}  // namespace si

extern "C" int si_host_sym_eig(int n, double* a, double* w) { return si::sym_eig(n, a, w); }
extern "C" int si_host_sym_eig_top(int n, const double* g, int m, double* w_top, double* V) {
  return si::sym_eig_top(n, g, m, w_top, V);
}
