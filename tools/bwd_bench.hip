// Timing of the reverse-sweep GEMMs (kernels_bwd.hip) one by one on the cfg2 layer shapes (development harness, not
// shipped).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I subspaceinference.jl_amd/csrc tools/bwd_bench.hip -o bwd_bench
// Usage: bwd_bench [out in B]    (defaults 960 960 100000)
#define SI_BWD_DEBUG_KNOB
#include <cstdlib>
#include "../subspaceinference.jl_amd/csrc/kernels_bwd.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
namespace si {
int32_t fail(Ctx*, int32_t c, const std::string&) { return c; }
ProfScope::ProfScope(Ctx*, int, double, double) {}
ProfScope::~ProfScope() {}
void launch_mul_dact(hipStream_t, const double*, const double*, int64_t, int, double*) {}   // extra activations: not timed here
}
using namespace si;

template <typename F>
static float best_ms(F f, int reps = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0, 0);
    f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms);
  }
  return best;
}

int main(int argc, char** argv) {
  const int out = argc > 1 ? atoi(argv[1]) : 960, in = argc > 2 ? atoi(argv[2]) : 960;
  const int64_t B = argc > 3 ? atoll(argv[3]) : 100000;
  const int dbg = argc > 4 ? atoi(argv[4]) : 0;
  hipMemcpyToSymbol(HIP_SYMBOL(si::si_bwd_dbg), &dbg, sizeof(int));
  if (dbg) printf("debug knob %d (results are wrong by design)\n", dbg);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  std::vector<double> hD((size_t)out * B), hH((size_t)in * B), hW((size_t)out * in);
  uint64_t s = 999;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((double)(s >> 11) / 9007199254740992.0) - 0.5; };
  for (auto& v : hD) v = rnd();
  for (auto& v : hH) v = rnd();
  for (auto& v : hW) v = rnd() * 0.1;
  double *dD, *dH, *dW, *dPart, *dG, *dDp, *dDb;
  hipMalloc(&dDb, (size_t)out * 8);
  hipMalloc(&dD, hD.size() * 8); hipMalloc(&dH, hH.size() * 8); hipMalloc(&dW, hW.size() * 8);
  hipMalloc(&dPart, std::max<size_t>(backward_weight_part_elems(out, in, B, ncu), (size_t)(plan_dw(out, in, B, ncu).nsplit + 1) * out * in) * 8); hipMalloc(&dG, (size_t)out * in * 8); hipMalloc(&dDp, (size_t)in * B * 8);
  hipMemcpy(dD, hD.data(), hD.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dH, hH.data(), hH.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice);
  const double flops = 2.0 * out * in * (double)B;
  {
    const DwPlan p = plan_dw(out, in, B, ncu);
    printf("out %d in %d B %lld  CUs %d  dW: %d x %d tiles, nsplit %d ksplit %lld\n", out, in, (long long)B, ncu, p.bm, p.bn, p.nsplit, (long long)p.ks);
  }
  float ms = best_ms([&] { launch_backward_weight(0, dD, dH, dPart, out, in, B, ncu, dG, dDb); });
  printf("dW   gemm + reduce     %8.3f ms  %6.2f TFLOP/s\n", ms, flops / (ms * 1e-3) / 1e12);
  // check dW against a host dot product on a few entries
  {
    std::vector<double> g((size_t)out * in);
    hipMemcpy(g.data(), dG, g.size() * 8, hipMemcpyDeviceToHost);
    double maxrel = 0;
    for (int t = 0; t < 24; ++t) {
      const int i = (int)((t * 7919u) % out), j = (int)((t * 104729u) % in);
      double ref = 0;
      for (int64_t b = 0; b < B; ++b) ref += hD[i + (size_t)out * b] * hH[j + (size_t)in * b];
      maxrel = fmax(maxrel, fabs(g[i + (size_t)out * j] - ref) / (fabs(ref) + 1e-9));
    }
    printf("dW   max rel err vs host on 24 entries: %.2e\n", maxrel);
    std::vector<double> db((size_t)out);
    hipMemcpy(db.data(), dDb, db.size() * 8, hipMemcpyDeviceToHost);
    double maxabs = 0;
    for (int t = 0; t < 12; ++t) {
      const int i = (int)((t * 7919u) % out);
      double ref = 0;
      for (int64_t b = 0; b < B; ++b) ref += hD[i + (size_t)out * b];
      maxabs = fmax(maxabs, fabs(db[i] - ref));
    }
    printf("db   max abs err vs host on 12 rows (sums of %lld values in [-0.5, 0.5)): %.2e\n", (long long)B, maxabs);
  }
  ms = best_ms([&] { launch_backward_data(0, dW, dD, dH, dDp, out, in, B, SI_ACT_RELU); });
  printf("dX   W' * Delta .* act' %8.3f ms  %6.2f TFLOP/s\n", ms, flops / (ms * 1e-3) / 1e12);
  {
    std::vector<double> dp((size_t)in * 64);
    hipMemcpy(dp.data(), dDp + (size_t)in * (B - 64), dp.size() * 8, hipMemcpyDeviceToHost);
    double maxrel = 0;
    for (int t = 0; t < 24; ++t) {
      const int j = (int)((t * 7919u) % in);
      const int64_t b = B - 64 + (t % 64);
      double ref = 0;
      for (int i = 0; i < out; ++i) ref += hW[i + (size_t)out * j] * hD[i + (size_t)out * b];
      ref *= hH[j + (size_t)in * b] > 0 ? 1.0 : 0.0;
      maxrel = fmax(maxrel, fabs(dp[j + (size_t)in * (b - (B - 64))] - ref) / (fabs(ref) + 1e-9));
    }
    printf("dX   max rel err vs host on 24 entries: %.2e\n", maxrel);
  }
  return 0;
}
