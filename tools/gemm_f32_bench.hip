// Tile-shape sweep for the fp32 MFMA dense kernel (kernels_gemm_f32.hip) on the cfg2 layer shapes -- development harness.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I subspaceinference.jl_amd/csrc tools/gemm_f32_bench.hip -o tools/bin/gemm_f32_bench
// Run:   gemm_f32_bench [out] [in] [B] [variant mask] [fused 0/1]
// (the debug knobs / alternative kernels behind profiles/r04_f32_kernel_experiments.log were development states of
//  kernels_gemm_f32.hip and are not kept)
#define SI_GEMM_F32_NO_DISPATCH
#include "../subspaceinference.jl_amd/csrc/kernels_gemm_f32.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
namespace si {
int32_t fail(Ctx*, int32_t c, const std::string&) { return c; }
ProfScope::ProfScope(Ctx*, int, double, double) {}
ProfScope::~ProfScope() {}
void launch_sse_final(hipStream_t, const double*, int, double*, int) {}
}
using namespace si;

// naive reference: one thread per output, fp64 accumulation of the fp32 operands
__global__ void ref_kernel(const float* W, const float* bias, const float* X, double* Y, int out, int in, int64_t B, int64_t b0, int nb) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)out * nb) return;
  const int i = (int)(idx % out);
  const int64_t b = b0 + idx / out;
  double s = 0.0;
  for (int k = 0; k < in; ++k) s += (double)W[i + (int64_t)out * k] * (double)X[k + (int64_t)in * b];
  s += (double)bias[i];
  Y[idx] = s > 0.0 ? s : 0.0;
}

typedef void (*launch_fn)(hipStream_t, const float*, const float*, const float*, float*, int32_t, int32_t, int64_t, int32_t, const FuseArgsF32&);
struct Variant { const char* name; launch_fn plain; launch_fn fused; int bm, wm; };
#define V(BM, BN, WM, WN, NB, MINW) {#BM "x" #BN " " #WM "x" #WN " nb" #NB " w" #MINW, \
  [](hipStream_t st, const float* W, const float* b, const float* X, float* Y, int32_t o, int32_t i, int64_t B, int32_t a, const FuseArgsF32& fa) { launch_f32_dma<BM, BN, WM, WN, NB, MINW, false>(st, W, b, X, Y, o, i, B, a, fa); }, \
  [](hipStream_t st, const float* W, const float* b, const float* X, float* Y, int32_t o, int32_t i, int64_t B, int32_t a, const FuseArgsF32& fa) { launch_f32_dma<BM, BN, WM, WN, NB, MINW, true>(st, W, b, X, Y, o, i, B, a, fa); }, BM, WM}

#define VK(BM, BN, WM, WN, NB, MINW, BK) {#BM "x" #BN " " #WM "x" #WN " nb" #NB " w" #MINW " bk" #BK, \
  [](hipStream_t st, const float* W, const float* b, const float* X, float* Y, int32_t o, int32_t i, int64_t B, int32_t a, const FuseArgsF32& fa) { launch_f32_dma<BM, BN, WM, WN, NB, MINW, false, BK>(st, W, b, X, Y, o, i, B, a, fa); }, \
  [](hipStream_t st, const float* W, const float* b, const float* X, float* Y, int32_t o, int32_t i, int64_t B, int32_t a, const FuseArgsF32& fa) { launch_f32_dma<BM, BN, WM, WN, NB, MINW, true, BK>(st, W, b, X, Y, o, i, B, a, fa); }, BM, WM}

int main(int argc, char** argv) {
  const int out = argc > 1 ? atoi(argv[1]) : 960, in = argc > 2 ? atoi(argv[2]) : 960;
  const int64_t B = argc > 3 ? atoll(argv[3]) : 100000;
  const unsigned long mask = argc > 4 ? strtoul(argv[4], nullptr, 0) : ~0ul;
  const int fused = argc > 5 ? atoi(argv[5]) : 0;

  std::vector<float> hW((size_t)out * in), hb(out), hX((size_t)in * B), hWl(out);
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)(((double)(s >> 11) / 9007199254740992.0) - 0.5); };
  for (auto& v : hW) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  for (auto& v : hX) v = rnd();
  for (auto& v : hWl) v = rnd();
  float *dW, *db, *dX, *dY, *dWl; double *dRef, *dPart;
  hipMalloc(&dW, hW.size() * 4); hipMalloc(&db, hb.size() * 4); hipMalloc(&dX, hX.size() * 4); hipMalloc(&dWl, hWl.size() * 4);
  hipMalloc(&dY, (size_t)out * B * 4); hipMalloc(&dRef, (size_t)out * 4096 * 8); hipMalloc(&dPart, (size_t)64 * B * 8);
  hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dWl, hWl.data(), hWl.size() * 4, hipMemcpyHostToDevice);
  const int64_t nb = std::min<int64_t>(B, 4096), bref0 = B - nb;
  hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((int64_t)out * nb + 255) / 256)), dim3(256), 0, 0, dW, db, dX, dRef, out, in, B, bref0, (int)nb);
  std::vector<double> ref((size_t)out * nb);
  hipMemcpy(ref.data(), dRef, ref.size() * 8, hipMemcpyDeviceToHost);
  std::vector<Variant> vs = {
    V(192, 128, 2, 4, 3, 4), V(128, 128, 2, 4, 3, 4), V(192, 128, 2, 4, 4, 2), V(192, 256, 2, 4, 3, 2), V(192, 64, 2, 2, 3, 3),
    V(96, 128, 1, 4, 3, 3), V(192, 128, 2, 4, 3, 2), V(192, 256, 2, 8, 3, 4),
    VK(192, 128, 2, 4, 2, 4, 32), VK(128, 128, 2, 4, 2, 4, 32), VK(192, 128, 2, 4, 2, 4, 16), VK(192, 128, 2, 4, 3, 2, 32),
    VK(192, 128, 2, 4, 2, 6, 16), VK(128, 128, 2, 4, 2, 6, 16), VK(128, 128, 2, 4, 2, 4, 16), V(192, 128, 2, 4, 3, 4),
    VK(192, 256, 2, 4, 2, 4, 16), VK(192, 256, 2, 4, 2, 2, 16), VK(192, 128, 2, 4, 2, 4, 16),
  };


  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flops = 2.0 * out * in * (double)B + (fused ? 2.0 * out * (double)B : 0.0);
  std::vector<float> got((size_t)out * nb);
  std::vector<double> gotp((size_t)nb);
  for (size_t v = 0; v < vs.size(); ++v) {
    if (!((mask >> v) & 1)) continue;
    FuseArgsF32 fa;
    if (fused) { fa.Wlast = dWl; fa.out_last = 1; fa.part = dPart; }
    auto run = [&]() { (fused ? vs[v].fused : vs[v].plain)(0, dW, db, dX, fused ? nullptr : dY, out, in, B, SI_ACT_RELU, fa); };
    hipMemset(dY, 0, (size_t)out * B * 4);
    run();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess || hipGetLastError() != hipSuccess) { printf("%-24s launch failed: %s\n", vs[v].name, hipGetErrorString(e)); continue; }
    double maxd = 0, maxr = 0;
    if (!fused) {
      hipMemcpy(got.data(), dY + (size_t)out * bref0, got.size() * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < got.size(); ++i) { maxd = fmax(maxd, fabs((double)got[i] - ref[i])); maxr = fmax(maxr, fabs(ref[i])); }
    } else {  // head: sum over slots of the partials vs sum_i Wl[i] * ref[i][b]
      const int slots = (out + vs[v].bm - 1) / vs[v].bm * vs[v].wm;
      std::vector<double> allp((size_t)slots * B);
      hipMemcpy(allp.data(), dPart, allp.size() * 8, hipMemcpyDeviceToHost);
      for (int64_t b = 0; b < nb; ++b) {
        double sum = 0, r = 0;
        for (int sl = 0; sl < slots; ++sl) sum += allp[(size_t)sl * B + bref0 + b];
        for (int i = 0; i < out; ++i) r += (double)hWl[i] * ref[(size_t)b * out + i];
        maxd = fmax(maxd, fabs(sum - r)); maxr = fmax(maxr, fabs(r));
      }
    }
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0); run(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
    }
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 20; ++rep) run();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms20; hipEventElapsedTime(&ms20, e0, e1);
    printf("%-24s best %7.3f ms %6.1f TF | 20 back-to-back %7.3f ms %6.1f TF = %.3f of 157.3 | maxerr %.2e (scale %.2e)\n", vs[v].name, best,
           flops / (best * 1e-3) / 1e12, ms20 / 20, flops / (ms20 / 20 * 1e-3) / 1e12, flops / (ms20 / 20 * 1e-3) / 1e12 / 157.3, maxd, maxr);
    fflush(stdout);
  }
  return 0;
}
