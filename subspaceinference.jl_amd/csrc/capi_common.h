// Declarations shared by the translation units of the C ABI (capi.hip: context + construction; capi_infer.hip: inference set-up,
// density, gradient, predictive forward; capi_sample.hip: the RWMH samplers and the output map).  Round 5 split the 2 200-line
// capi.hip along these seams; nothing here is part of the public interface (include/subspace_hip.h).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "si_internal.h"

#ifdef SI_DEV_KNOBS   // development build: the library's own buffers through the guard-page allocator (guard_alloc.hip; SI_GUARD_ALLOC=end|begin)
namespace si {
hipError_t guard_malloc(void** out, size_t bytes);
hipError_t guard_free(void* p);
}
#define hipMalloc(p, n) si::guard_malloc((void**)(p), (n))
#define hipFree(p) si::guard_free((void*)(p))
#endif

namespace si {

template <typename T>
static inline hipError_t dev_alloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
}
template <typename T>
static inline void dev_free(T*& p) {
  if (p) (void)hipFree((void*)p);
  p = nullptr;
}

void free_infer(Ctx* c);   // capi.hip: everything si_infer_setup allocated

}  // namespace si

#define CHECK_CTX(ctx) \
  if (!(ctx)) return SI_ERR_INVALID
#define BIND(ctx) SI_HIP(ctx, hipSetDevice((ctx)->device))

extern "C" {   // (defined inside the extern "C" blocks of their translation units; not exported by include/subspace_hip.h)
// capi_infer.hip
int32_t ensure_chains(si_ctx* ctx, int32_t C);   // forward workspace + sampler state for C chains
int32_t eval_density(si_ctx* ctx, int c0, int nc, const double** yhat_out);   // d_zprop[:, c0 .. c0+nc) -> d_sse
int32_t eval_density_all(si_ctx* ctx, int C);
double mvnormal_c0(double d, double sigma);
double prior_c0(const si_ctx* ctx);
void fused_fill_program(const si_ctx* ctx, si::ChainFusedPlan& fp);
// capi_sample.hip
void free_wstream(si_ctx* ctx);
}
