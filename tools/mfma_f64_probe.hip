// Probe of v_mfma_f64_16x16x4_f64 on gfx950: (1) discovers the C/D lane/register -> (row, col) map and checks
// the A/B operand maps the kernels assume; (2) measures issue cycles per MFMA (independent and dependent
// accumulators).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/, double* out /*64 lanes x 4*/) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];   // A[i = l&15][k = l>>4]
  const double b = B[(l >> 4) * 16 + (l & 15)];  // B[k = l>>4][j = l&15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void rate_kernel(double* out, long long* cyc, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 0.001 + 1.0, b = 0.5 - threadIdx.x * 0.002;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// sustained-rate probe: operands either constant or random per lane, static register indices, per-block stamps.
// s_memtime = shader-clock ticks (guide), s_memrealtime = 100 MHz.
template <bool RANDOM>
__global__ void clock_kernel(double* out, long long* stamps, const double* rnd, int iters) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = RANDOM ? rnd[(threadIdx.x * 8 + i + 64 * blockIdx.x) & 4095] : 1.0 + 0.25 * i;
    b[i] = RANDOM ? rnd[(threadIdx.x * 8 + 4 + i + 32 * blockIdx.x) & 4095] : 0.5 - 0.125 * i;
  }
  long long t0 = __builtin_amdgcn_s_memtime();
  long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i + u) & 3], acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  std::vector<double> A(64), B(64), D(256, 0.0), out(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 4 + k;           // distinct
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 100 + 7 * k + 31 * j;    // asymmetric
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) D[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dO;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dO, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dO);
  hipMemcpy(out.data(), dO, 2048, hipMemcpyDeviceToHost);
  int ok_guide = 0, ok_f32map = 0, found = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const double v = out[l * 4 + r];
    if (v == D[((l >> 4) + 4 * r) * 16 + (l & 15)]) ok_guide++;
    if (v == D[((l >> 4) * 4 + r) * 16 + (l & 15)]) ok_f32map++;
    for (int e = 0; e < 256; ++e) if (D[e] == v) { found++; if (l < 2 || l == 17 || l == 63) printf("lane %d reg %d -> row %d col %d\n", l, r, e / 16, e % 16); break; }
  }
  printf("LAYOUT: guide map row=(l>>4)+4r col=l&15 matches %d/256; f32 map matches %d/256; values located %d/256\n", ok_guide, ok_f32map, found);

  int ncu = 256; hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); ncu = p.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, ncu, p.clockRate);
  double* dOut; long long* dCyc; hipMalloc(&dOut, sizeof(double) * 4096 * 256); hipMalloc(&dCyc, 8 * 4096);
  const int iters = 20000;
  auto run = [&](auto kern, int nacc, int blocks, int threads, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dOut, dCyc, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dOut, dCyc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, dCyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * nacc;
    const double waves = (double)blocks * threads / 64;
    printf("%s: blocks %d x %d thr: %.1f memtime-ticks/MFMA per wave, %.3f ms, %.2f TFLOP/s\n", name, blocks, threads,
           (double)c / nm, ms, nm * waves * 2048.0 / (ms * 1e-3) / 1e12);
  };
  run(rate_kernel<1>, 1, 1, 64, "1 wave, 1 dependent acc");
  run(rate_kernel<4>, 4, 1, 64, "1 wave, 4 independent acc");
  run(rate_kernel<8>, 8, 1, 64, "1 wave, 8 independent acc");
  run(rate_kernel<8>, 8, ncu, 256, "1 wave/SIMD all CUs, 8 acc");
  run(rate_kernel<8>, 8, ncu * 2, 256, "2 waves/SIMD all CUs, 8 acc");
  run(rate_kernel<4>, 4, ncu * 4, 256, "4 waves/SIMD all CUs, 4 acc");
  {
    std::vector<double> hr(4096);
    unsigned long long st = 88172645463325252ull;
    for (auto& v : hr) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
    double* dR; hipMalloc(&dR, 4096 * 8); hipMemcpy(dR, hr.data(), 4096 * 8, hipMemcpyHostToDevice);
    long long* dS; hipMalloc(&dS, 16 * 4096);
    auto sustained = [&](auto kern, const char* name, int wgs) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int blocks = ncu * wgs, its = 400000 / wgs;   // ~0.15-0.25 s of back-to-back MFMAs
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dOut, dS, dR, 1000);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dOut, dS, dR, its);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> hs(2 * blocks); hipMemcpy(hs.data(), dS, 16 * blocks, hipMemcpyDeviceToHost);
      std::vector<double> rt(blocks), cy(blocks);
      for (int i = 0; i < blocks; ++i) { rt[i] = hs[2 * i + 1] * 1e-5; cy[i] = (double)hs[2 * i] / ((double)its * 8); }
      std::sort(rt.begin(), rt.end()); std::sort(cy.begin(), cy.end());
      printf("%s, %d waves/SIMD: wall %.1f ms => %.2f TFLOP/s; per-block realtime ms min/med/max %.1f/%.1f/%.1f; memtime ticks per MFMA per wave min/med/max %.1f/%.1f/%.1f; memtime/realtime %.3f GHz\n",
             name, wgs, ms, (double)its * 8 * blocks * 4 * 2048.0 / (ms * 1e-3) / 1e12, rt[0], rt[blocks / 2], rt[blocks - 1],
             cy[0], cy[blocks / 2], cy[blocks - 1], (double)hs[0] / (double)hs[1] * 0.1);
    };
    sustained(clock_kernel<false>, "constant operands", 1);
    sustained(clock_kernel<true>, "random operands", 1);
    sustained(clock_kernel<false>, "constant operands", 2);
    sustained(clock_kernel<true>, "random operands", 2);
  }
  return 0;
}
