// Device helpers shared by the device-resident RWMH loops (kernels_chain.hip: one workgroup per chain; kernels_chain_grid.hip:
// a grid of workgroups per chain).  They reproduce, operation for operation, pieces of the launch-per-step kernels: the
// activation of the MFMA epilogues (kernels_gemm.hip apply_act), the 16-lane butterfly of the fused head and the
// shuffle-down wave sum of the SSE kernels (kernels_stream.hip wave_sum).
#pragma once
#ifdef __HIPCC_RTC__   // compiled at run time with chain_spec.inc: only the activation codes of include/subspace_hip.h are needed
enum { SI_ACT_IDENTITY = 0, SI_ACT_RELU = 1, SI_ACT_TANH = 2, SI_ACT_SIGMOID = 3 };   // (checked against the header in chain_spec_rtc.cpp)
typedef unsigned int uint32_t;   // (hiprtc keeps its fixed-width types in a namespace of its own)
typedef unsigned long uint64_t;
typedef long int64_t;
typedef unsigned long uintptr_t;
#else
#include "kernels_gemm.h"
#endif

namespace si {

__device__ __forceinline__ double chain_act(double v, int act) {
  switch (act) {
    case SI_ACT_RELU: return v > 0.0 ? v : 0.0;
    case SI_ACT_TANH: return tanh(v);
    case SI_ACT_SIGMOID: return 1.0 / (1.0 + exp(-v));
    default: return v;
  }
}

// x with its lanes rotated by N inside every row of 16 lanes (DPP row_ror): for a value that is already periodic with period
// 2N inside the row -- every step of a 16-lane xor butterfly that started at 8 -- this IS __shfl_xor(x, N, 16), without the
// LDS round trip of ds_bpermute
template <int N>
__device__ __forceinline__ double chain_row_ror(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x120 | N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x120 | N, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// lane i <- lane i + N of the same row of 16 (DPP row_shl); lanes whose source leaves the row keep their own value: they do
// not reach lane 0 of a shuffle-down tree
template <int N>
__device__ __forceinline__ double chain_row_shl(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)b, (int)(unsigned)b, 0x100 | N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), 0x100 | N, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// lane 0 of the result == lane 0 of `for (off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64)` (wave_sum of
// kernels_stream.hip / tail_sse_kernel): the two row-crossing steps as shuffles, the four in-row steps as DPP moves
__device__ __forceinline__ double chain_wave_sum(double v) {
  v += __shfl_down(v, 32, 64);
  v += __shfl_down(v, 16, 64);
  v += chain_row_shl<8>(v);
  v += chain_row_shl<4>(v);
  v += chain_row_shl<2>(v);
  v += chain_row_shl<1>(v);
  return v;
}

}  // namespace si
