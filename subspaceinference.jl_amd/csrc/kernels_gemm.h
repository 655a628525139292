// shared declarations of the fp64 MFMA dense kernels (kernels_gemm.hip, tools/gemm_bench.hip)
#pragma once
#include "si_internal.h"

namespace si {
typedef double d4 __attribute__((ext_vector_type(4)));

// The activations beyond identity / relu / tanh / sigmoid (include/subspace_hip.h).  They are NOT compiled into the MFMA
// kernels' epilogues: inlined at every unrolled call site they more than doubled the code of those kernels and cost the
// store-bound layer-1 kernel of cfg2 8 % (instruction fetch).  A layer with one of them runs its GEMM with the identity and is
// finished by one elementwise pass (launch_act_inplace / launch_mul_dact) -- all of them are increasing, so they also
// commute with the fused MaxPool.  Elementwise kernels use act_full / dact_full.
constexpr double SI_SELU_LAMBDA = 1.0507009873554805, SI_SELU_ALPHA = 1.6732632423543772;
__device__ __forceinline__ double act_extra(double v, int act) {
  switch (act) {
    case SI_ACT_LEAKYRELU: return v > 0.01 * v ? v : 0.01 * v;                       // max(a x, x)
    case SI_ACT_ELU: return v >= 0.0 ? v : exp(v) - 1.0;
    case SI_ACT_SOFTPLUS: return v > 0.0 ? v + log1p(exp(-v)) : log1p(exp(v));
    case SI_ACT_SELU: return SI_SELU_LAMBDA * (v > 0.0 ? v : SI_SELU_ALPHA * (exp(v) - 1.0));
    default: return v;
  }
}
// act'(x) expressed through the OUTPUT h = act(x) (all four are strictly increasing, so x -> h is invertible)
__device__ __forceinline__ double dact_extra(double h, int act) {
  switch (act) {
    case SI_ACT_LEAKYRELU: return h > 0.0 ? 1.0 : 0.01;
    case SI_ACT_ELU: return h >= 0.0 ? 1.0 : h + 1.0;                                 // exp(x) = h + 1
    case SI_ACT_SOFTPLUS: return 1.0 - exp(-h);                                       // sigmoid(x) = 1 - exp(-softplus(x))
    case SI_ACT_SELU: return h > 0.0 ? SI_SELU_LAMBDA : h + SI_SELU_LAMBDA * SI_SELU_ALPHA;
    default: return 1.0;
  }
}
__device__ __forceinline__ double act_full(double v, int act) {
  switch (act) {
    case SI_ACT_IDENTITY: return v;
    case SI_ACT_RELU: return v > 0.0 ? v : 0.0;
    case SI_ACT_TANH: return tanh(v);
    case SI_ACT_SIGMOID: return 1.0 / (1.0 + exp(-v));
    default: return act_extra(v, act);
  }
}
__device__ __forceinline__ double dact_full(double h, int act) {
  switch (act) {
    case SI_ACT_IDENTITY: return 1.0;
    case SI_ACT_RELU: return h > 0.0 ? 1.0 : 0.0;
    case SI_ACT_TANH: return 1.0 - h * h;
    case SI_ACT_SIGMOID: return h * (1.0 - h);
    default: return dact_extra(h, act);
  }
}
// the same in the layer's own precision (the fp32 reverse sweep: kernels_bwd_f32.hip and the dX epilogue of kernels_gemm_f32.hip)
__device__ __forceinline__ float dact_f32(float h, int act) {
  switch (act) {
    case SI_ACT_IDENTITY: return 1.0f;
    case SI_ACT_RELU: return h > 0.0f ? 1.0f : 0.0f;
    case SI_ACT_TANH: return 1.0f - h * h;
    case SI_ACT_SIGMOID: return h * (1.0f - h);
    default: return (float)dact_extra((double)h, act);
  }
}
inline bool act_is_extra(int act) { return act >= SI_ACT_LEAKYRELU; }
}
