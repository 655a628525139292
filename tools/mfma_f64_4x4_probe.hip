// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result lane layout and issue rate.  Development tool (not part of the
// library): the ragged last column tile of the Gram kernel uses this instruction, and the guide has no layout table for it.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_f64_4x4_probe tools/mfma_f64_4x4_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void probe(double* out /* [64 la][64 ld] */) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    const double a = lane == la ? 1.0 : 0.0;
    const double b = (double)(lane + 1);
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[la * 64 + lane] = d;
  }
}

__global__ __launch_bounds__(256) void rate(double* out, int iters, int which) {
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    if (which == 0) {
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    } else if (which == 1) {
      e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e1, 0, 0, 0);
    } else if (which == 2) {   // alternating shapes: big small big small
      e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e0, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e1, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    } else {                   // grouped: big big small small
      e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + e0[0] + e0[1] + e0[2] + e0[3] + e1[0] + e1[1] + e1[2] + e1[3];
}

int main() {
  double* d;
  hipMalloc(&d, 64 * 64 * sizeof(double));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<double> h(64 * 64);
  hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  printf("one-hot A at lane la: result lanes ld with the B lane (value - 1) that met it\n");
  for (int la = 0; la < 64; ++la) {
    printf("la %2d:", la);
    for (int ld = 0; ld < 64; ++ld)
      if (h[la * 64 + ld] != 0.0) printf("  ld %2d <- lb %2d", ld, (int)h[la * 64 + ld] - 1);
    printf("\n");
  }
  // issue rate: 1024 blocks x 4 waves, dependent chains of 4 (4x4x4) / 2 (16x16x4) accumulators
  double* o;
  hipMalloc(&o, 1024 * 256 * sizeof(double));
  for (int which = 0; which < 4; ++which) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, o, 100, which);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, o, iters, which);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (which >= 2) {
      // per iteration and wave: 2 x 16x16x4 + 2 x 4x4x4_4b; 4 waves per SIMD
      printf("%s: %.3f ms = %.1f cycles per (2 big + 2 small) per SIMD-wave slot at 2.4 GHz\n",
             which == 2 ? "alternating big/small" : "grouped big big small small", ms, ms * 1e-3 * 2.4e9 / (4.0 * iters));
      continue;
    }
    const double n_inst = (double)1024 * 4 * iters * (which == 0 ? 4 : 2);
    const double macs = n_inst * (which == 0 ? 256.0 : 1024.0);
    printf("%s: %.3f ms, %.2f TFLOP/s, %.1f cycles per instruction per SIMD at 2.4 GHz (1024 SIMDs)\n",
           which == 0 ? "v_mfma_f64_4x4x4_4b" : "v_mfma_f64_16x16x4", ms, 2 * macs / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (n_inst / 1024.0));
  }
  return 0;
}
