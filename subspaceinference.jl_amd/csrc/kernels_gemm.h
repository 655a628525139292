// shared declarations of the fp64 MFMA dense kernels (kernels_gemm.hip, tools/gemm_bench.hip)
#pragma once
#include "si_internal.h"

namespace si {
typedef double d4 __attribute__((ext_vector_type(4)));
}
