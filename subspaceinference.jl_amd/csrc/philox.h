// Philox4x32-10 counter RNG, device side.  Bit-for-bit twin of oracle/philox.py (stream definition there).
// Replaces Julia's global MersenneTwister draws inside AdvancedMH's RWMH (reference
// src/space_inference.jl:113-116): `rand(rng, proposal)` and `randexp(rng)`.
#pragma once
#ifndef __HIPCC_RTC__   // (hiprtc brings the runtime and the fixed-width types itself)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace si {

struct Philox4 {
  uint32_t v[4];
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                  uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

__host__ __device__ inline double u53(uint32_t hi, uint32_t lo) {
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  return ((double)(v >> 11) + 0.5) * 0x1.0p-53;
}

// counter = (step lo, step hi, chain, purpose<<24 | block); key = (seed lo, seed hi)
__host__ __device__ inline Philox4 philox_draw(uint64_t seed, uint32_t chain, uint64_t step,
                                                uint32_t purpose, uint32_t block) {
  return philox4x32_10((uint32_t)step, (uint32_t)(step >> 32), chain,
                       ((purpose & 0xFFu) << 24) | (block & 0xFFFFFFu), (uint32_t)seed,
                       (uint32_t)(seed >> 32));
}

// two standard normals (components 2*block, 2*block+1 of the proposal noise of `step`)
__device__ inline void philox_normal2(uint64_t seed, uint32_t chain, uint64_t step, uint32_t block,
                                      double& n0, double& n1) {
  const Philox4 x = philox_draw(seed, chain, step, 0u, block);
  const double u1 = u53(x.v[1], x.v[0]);
  const double u2 = u53(x.v[3], x.v[2]);
  const double r = sqrt(-2.0 * log(u1));
  const double t = (2.0 * 3.14159265358979323846) * u2;
  n0 = r * cos(t);
  n1 = r * sin(t);
}

__device__ inline double philox_randexp(uint64_t seed, uint32_t chain, uint64_t step) {
  const Philox4 x = philox_draw(seed, chain, step, 1u, 0u);
  return -log(u53(x.v[1], x.v[0]));
}

}  // namespace si
