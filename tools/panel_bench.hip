// The panel kernel for a wide first layer (csrc/kernels_gemm_panel.hip) against dense_f64_kernel on the same operands: bits
// of the whole output, time per launch alone and in a back-to-back queue (development harness, not shipped).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I subspaceinference.jl_amd/csrc -I include tools/panel_bench.hip -o tools/bin/panel_bench
// Usage: panel_bench [out 960] [in 128] [B 100000] [act 1] [grid list e.g. 512,256,1024]
#define SI_GEMM_NO_DISPATCH
#include "../subspaceinference.jl_amd/csrc/kernels_gemm.hip"
#include "../subspaceinference.jl_amd/csrc/kernels_gemm_panel.hip"
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <unistd.h>
#include <vector>
namespace si {
int32_t fail(Ctx*, int32_t c, const std::string&) { return c; }
ProfScope::ProfScope(Ctx*, int, double, double) {}
ProfScope::~ProfScope() {}
}
using namespace si;

int main(int argc, char** argv) {
  const int out = argc > 1 ? atoi(argv[1]) : 960, in = argc > 2 ? atoi(argv[2]) : 128;
  const int64_t B = argc > 3 ? atoll(argv[3]) : 100000;
  const int act = argc > 4 ? atoi(argv[4]) : 1;
  std::vector<int> nrts;
  {
    const char* p = argc > 5 ? argv[5] : "512";
    while (*p) { nrts.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
  }
  std::vector<double> hW((size_t)out * in), hb(out), hX((size_t)in * B);
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((double)(s >> 11) / 9007199254740992.0) - 0.5; };
  for (auto& v : hW) v = rnd() * 0.1;
  for (auto& v : hb) v = rnd();
  for (auto& v : hX) v = rnd();
  double *dW, *db, *dX, *dY, *dRef;
  const size_t ybytes = (size_t)out * B * 8;
  if (hipMalloc(&dW, hW.size() * 8) || hipMalloc(&db, hb.size() * 8) || hipMalloc(&dX, hX.size() * 8) || hipMalloc(&dY, ybytes) || hipMalloc(&dRef, ybytes)) return 1;
  (void)hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(db, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dX, hX.data(), hX.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const double flops = 2.0 * out * in * (double)B;
  std::vector<double> ref((size_t)out * B), got((size_t)out * B);
  auto run = [&](const char* name, auto&& fn, double* dst, bool check) {
    (void)hipMemset(dst, 0xff, ybytes);
    fn(dst);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); exit(1); }
    (void)hipMemcpy(check ? got.data() : ref.data(), dst, ybytes, hipMemcpyDeviceToHost);
    size_t bad = 0;
    if (check) for (size_t i = 0; i < ref.size(); ++i) bad += std::memcmp(&ref[i], &got[i], 8) != 0;
    float best = 1e30f, ms10;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0, 0); fn(dst); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
    }
    (void)hipEventRecord(e0, 0);
    for (int rep = 0; rep < 10; ++rep) fn(dst);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms10, e0, e1);
    {  // "cool": a 3 ms pause in front of every launch -- the chip is not held at its power limit by the launches before it
      std::vector<float> t;
      for (int rep = 0; rep < 25; ++rep) {
        usleep(3000);
        (void)hipEventRecord(e0, 0); fn(dst); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
      }
      std::sort(t.begin(), t.end());
      printf("[cool: min %.4f med %.4f ms] ", t[0], t[t.size() / 2]);
    }
    printf("%-22s alone %7.4f ms  queued %7.4f ms = %6.2f TFLOP/s  elements that differ from dense_f64_kernel: %zu\n", name, best, ms10 / 10,
           flops / (ms10 / 10 * 1e-3) / 1e12, bad);
#if defined(SI_PANEL_KNOB) && (SI_PANEL_KNOB & 8)
    if (check) {   // in-kernel clock and cycles per unit of the LAST launch (the queued ones)
      std::vector<long long> st(4 * 2048);
      (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(si::si_panel_stamps), st.size() * 8);
      std::vector<double> cyc, clk, pro, loop;
      long long first = 1ll << 62, lastentry = 0, lastend = 0, firstend = 1ll << 62;
      for (int g = 0; g < 2048; ++g) {
        const long long* t = &st[4 * g];
        if (t[3] <= t[2]) continue;
        cyc.push_back((double)t[0]); clk.push_back((double)t[0] / (double)(t[3] - t[2]) * 0.1);
        pro.push_back((t[2] - t[1]) * 0.01); loop.push_back((t[3] - t[2]) * 0.01);
        first = std::min(first, t[1]); lastentry = std::max(lastentry, t[1]); lastend = std::max(lastend, t[3]); firstend = std::min(firstend, t[3]);
      }
      std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end()); std::sort(pro.begin(), pro.end()); std::sort(loop.begin(), loop.end());
      {
        double xs[8] = {0}, xc[8] = {0}, xmx[8] = {0}; int xn[8] = {0};
        for (int g = 0; g < 2048; ++g) {
          const long long* t = &st[4 * g];
          if (t[3] <= t[2]) continue;
          xs[g & 7] += (t[3] - t[2]) * 0.01; xc[g & 7] += (double)t[0] / (double)(t[3] - t[2]) * 0.1; xn[g & 7]++;
          xmx[g & 7] = std::max(xmx[g & 7], (t[3] - t[2]) * 0.01);
        }
        printf("   per XCD (workgroup index mod 8): loop us mean / max, clock GHz:");
        for (int x = 0; x < 8; ++x) if (xn[x]) printf("  %.0f/%.0f %.2f", xs[x] / xn[x], xmx[x], xc[x] / xn[x]);
        printf("\n");
      }
      const double units = (double)((B + 127) / 128) * ((out + 15) / 16) / (double)cyc.size();
      if (!cyc.empty())
        printf("   in-kernel: clock med %.3f GHz, unit loop med %.0f cycles = %.0f per unit (8192 = four waves' MFMAs back to back)\n"
               "   us: workgroup entries spread %.1f, prologue med %.1f max %.1f, loop med %.1f max %.1f, first entry -> first end %.1f -> last end %.1f\n",
               clk[clk.size() / 2], cyc[cyc.size() / 2], cyc[cyc.size() / 2] / units, (lastentry - first) * 0.01, pro[pro.size() / 2], pro.back(),
               loop[loop.size() / 2], loop.back(), (firstend - first) * 0.01, (lastend - first) * 0.01);
    }
#endif
    fflush(stdout);
  };
  run("dense_f64 96x128", [&](double* y) { launch_dense_cfg<96, 128, 2, 4, 4>(0, dW, db, dX, y, out, in, B, act); }, dRef, false);
  for (int nrt : nrts) {
    char name[64];
    snprintf(name, sizeof name, "panel grid %d", nrt);
    run(name, [&](double* y) { if (!launch_dense_f64_panel(0, dW, db, dX, y, out, in, B, act, nrt)) { printf("outside the class\n"); exit(2); } }, dY, true);
  }
  return 0;
}
