// Can fp64 VALU FMAs ride along with fp64 MFMAs?  Loop of [1 v_mfma_f64_16x16x4_f64 + NV independent v_fma_f64] per
// wave, register operands only; reports time vs the MFMA-only loop.  If NV FMAs per MFMA are free, a hybrid GEMM could
// exceed the 78.6 TFLOP/s matrix peak (each v_fma_f64 wave-instruction = 128 flop, each MFMA = 2048 flop).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256, 4) void k(double* out, int iters) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001 + i;
  const double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3, c = 1.0000001, d = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[(m * NV + j) & 15] = __builtin_fma(v[(m * NV + j) & 15], c, d);
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
void run(double* out, int ncu, int wgs) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NV>, dim3(ncu * wgs), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<NV>, dim3(ncu * wgs), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)ncu * wgs * 4, nm = (double)iters * 4;
  const double tf_m = nm * waves * 2048 / (ms * 1e-3) / 1e12, tf_v = nm * NV * waves * 128 / (ms * 1e-3) / 1e12;
  printf("NV=%2d waves/SIMD=%d: %.3f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TF fp64\n", NV, wgs, ms, tf_m, tf_v, tf_m + tf_v);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  double* out; hipMalloc(&out, sizeof(double) * 256 * p.multiProcessorCount * 4);
  for (int wgs = 1; wgs <= 2; ++wgs) {
    run<0>(out, p.multiProcessorCount, wgs);
    run<2>(out, p.multiProcessorCount, wgs);
    run<4>(out, p.multiProcessorCount, wgs);
    run<8>(out, p.multiProcessorCount, wgs);
    run<12>(out, p.multiProcessorCount, wgs);
    run<16>(out, p.multiProcessorCount, wgs);
  }
  return 0;
}
