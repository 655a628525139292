// Tile-shape sweep for the fp64 MFMA dense kernel on the cfg2 layer shapes (development harness, not shipped).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I subspaceinference.jl_amd/csrc tools/gemm_bench.hip -o gemm_bench
#define SI_GEMM_NO_DISPATCH
#define SI_GEMM_DEBUG_KNOB
#include "../subspaceinference.jl_amd/csrc/kernels_gemm.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
namespace si {
int32_t fail(Ctx*, int32_t c, const std::string&) { return c; }
ProfScope::ProfScope(Ctx*, int, double, double) {}
ProfScope::~ProfScope() {}
}
using namespace si;

typedef void (*launch_fn)(hipStream_t, const double*, const double*, const double*, double*, int32_t, int32_t, int64_t, int32_t);
struct Variant { const char* name; launch_fn fn; };

int main(int argc, char** argv) {
  const int out = argc > 1 ? atoi(argv[1]) : 960, in = argc > 2 ? atoi(argv[2]) : 960;
  const int64_t B = argc > 3 ? atoll(argv[3]) : 100000;
  const int dbg = argc > 5 ? atoi(argv[5]) : 0;
  hipMemcpyToSymbol(HIP_SYMBOL(si::si_gemm_dbg), &dbg, sizeof(int));
  si::si_gemm_lds_floor = argc > 6 ? (size_t)atol(argv[6]) : 0;

  printf("debug knob %d lds floor %zu\n", dbg, si::si_gemm_lds_floor);
  const unsigned long mask = argc > 4 ? strtoul(argv[4], nullptr, 0) : ~0ul;  // bit v selects variant v (0 always runs)
  std::vector<double> hW((size_t)out * in), hb(out), hX((size_t)in * B);
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return ((double)(s >> 11) / 9007199254740992.0) - 0.5; };
  for (auto& v : hW) v = rnd() * 0.1;
  for (auto& v : hb) v = rnd();
  for (auto& v : hX) v = rnd();
  double *dW, *db, *dX, *dY, *dRef;
  hipMalloc(&dW, hW.size() * 8); hipMalloc(&db, hb.size() * 8); hipMalloc(&dX, hX.size() * 8);
  hipMalloc(&dY, (size_t)out * B * 8); hipMalloc(&dRef, (size_t)out * B * 8);
  hipMemcpy(dW, hW.data(), hW.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dX, hX.data(), hX.size() * 8, hipMemcpyHostToDevice);
  std::vector<Variant> vs = {
    {"128x128 2x2 w2 (baseline)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<128, 128, 2, 2, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"160x128 2x2 w2", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<160, 128, 2, 2, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"192x128 2x2 w1", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<192, 128, 2, 2, 1>(st, W, b, X, Y, o, i, B, a); }},
    {"192x128 4x2 w2 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<192, 128, 4, 2, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"192x128 2x4 w2 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<192, 128, 2, 4, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"160x128 2x4 w4 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<160, 128, 2, 4, 4>(st, W, b, X, Y, o, i, B, a); }},
    {"192x256 2x4 w2 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<192, 256, 2, 4, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"160x256 2x4 w2 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<160, 256, 2, 4, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"192x192 2x4 w2 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<192, 192, 2, 4, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"128x256 2x4 w2 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<128, 256, 2, 4, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"96x128 2x2 w2", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<96, 128, 2, 2, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"64x128 2x2 w2", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<64, 128, 2, 2, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"32x128 1x4 w2", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<32, 128, 1, 4, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"96x128 2x4 w4 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<96, 128, 2, 4, 4>(st, W, b, X, Y, o, i, B, a); }},
    {"128x128 2x4 w4 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<128, 128, 2, 4, 4>(st, W, b, X, Y, o, i, B, a); }},
    {"128x128 4x2 w4 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<128, 128, 4, 2, 4>(st, W, b, X, Y, o, i, B, a); }},
    {"64x128 2x4 w4 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<64, 128, 2, 4, 4>(st, W, b, X, Y, o, i, B, a); }},
    {"96x256 2x4 w2 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<96, 256, 2, 4, 2>(st, W, b, X, Y, o, i, B, a); }},
    {"192x128 4x2 w3 (8 waves)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<192, 128, 4, 2, 3>(st, W, b, X, Y, o, i, B, a); }},
    {"64x64 2x2 w4", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<64, 64, 2, 2, 4>(st, W, b, X, Y, o, i, B, a); }},
    {"64x64 2x4 w6 (8 waves, 3 WG/CU)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<64, 64, 2, 4, 6>(st, W, b, X, Y, o, i, B, a); }},
    {"96x64 2x4 w6 (8 waves, 3 WG/CU)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<96, 64, 2, 4, 6>(st, W, b, X, Y, o, i, B, a); }},
    {"64x64 2x4 w8 (8 waves, 4 WG/CU)", [](hipStream_t st, const double* W, const double* b, const double* X, double* Y, int32_t o, int32_t i, int64_t B, int32_t a) { launch_dense_cfg<64, 64, 2, 4, 8>(st, W, b, X, Y, o, i, B, a); }},
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> ref((size_t)out * 4096), got((size_t)out * 4096);
  const double flops = 2.0 * out * in * (double)B;
  for (size_t v = 0; v < vs.size(); ++v) {
    if (v > 0 && !((mask >> v) & 1)) continue;
    hipMemset(dY, 0, (size_t)out * B * 8);
    vs[v].fn(0, dW, db, dX, dY, out, in, B, SI_ACT_RELU);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess || hipGetLastError() != hipSuccess) { printf("%-28s launch failed: %s\n", vs[v].name, hipGetErrorString(e)); continue; }
    // compare the LAST 4096 columns (covers the ragged edge) with the baseline variant
    const size_t off = (size_t)out * (B > 4096 ? B - 4096 : 0), cnt = (size_t)out * (B > 4096 ? 4096 : B);
    hipMemcpy(v == 0 ? ref.data() : got.data(), dY + off, cnt * 8, hipMemcpyDeviceToHost);
    double maxd = 0;
    if (v > 0) for (size_t i = 0; i < cnt; ++i) maxd = fmax(maxd, fabs(got[i] - ref[i]));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      vs[v].fn(0, dW, db, dX, dY, out, in, B, SI_ACT_RELU);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
    }
    {  // ten launches back to back under ONE event pair: is the per-launch cost paid again without a host gap?
      hipEventRecord(e0, 0);
      for (int rep = 0; rep < 10; ++rep) vs[v].fn(0, dW, db, dX, dY, out, in, B, SI_ACT_RELU);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms10; hipEventElapsedTime(&ms10, e0, e1);
      printf("[10 back-to-back: %.3f ms each] ", ms10 / 10);
    }
    printf("%-28s %8.3f ms  %6.2f TFLOP/s  maxdiff-vs-baseline %.3e", vs[v].name, best, flops / (best * 1e-3) / 1e12, maxd);
    {
      std::vector<long long> st(4 * 16384);
      hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(si::si_gemm_stamps), st.size() * 8);
      std::vector<double> pro, loop, epi;
      for (int b = 0; b < 2048; ++b) { const long long* t = &st[4 * b]; if (t[3] > t[0] && t[0] > 0) { pro.push_back(t[1] - t[0]); loop.push_back(t[2] - t[1]); epi.push_back(t[3] - t[2]); } }
      if (!pro.empty()) { std::sort(pro.begin(), pro.end()); std::sort(loop.begin(), loop.end()); std::sort(epi.begin(), epi.end());
        printf("  | cycles med: prologue %.0f loop %.0f epilogue %.0f (n=%zu)", pro[pro.size()/2], loop[loop.size()/2], epi[epi.size()/2], pro.size()); }
      std::fill(st.begin(), st.end(), 0); hipMemcpyToSymbol(HIP_SYMBOL(si::si_gemm_stamps), st.data(), st.size() * 8);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
