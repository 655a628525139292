// Where a transition of the device-resident RWMH loop (kernels_chain.hip) spends its cycles: README-toy shaped chain, phase
// stamps of chain 0.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSI_CHAIN_STAMPS -I subspaceinference.jl_amd/csrc tools/chain_bench.hip -o tools/bin/chain_bench
#include "../subspaceinference.jl_amd/csrc/kernels_chain.hip"
#include <cstdio>
#include <vector>
namespace si {
int32_t fail(Ctx*, int32_t c, const std::string&) { return c; }
ProfScope::ProfScope(Ctx*, int, double, double) {}
ProfScope::~ProfScope() {}
}
using namespace si;
int main(int argc, char** argv) {
  const int dims[4] = {10, 20, 20, 2}, B = 100, M = 3;
  const int64_t itr = argc > 1 ? atoll(argv[1]) : 2000;
  const int nch = argc > 2 ? atoi(argv[2]) : 1;
  ChainLoopArgs a{};
  int off = 0;
  for (int l = 0; l < 3; ++l) {
    a.lay[l].kind = 0; a.lay[l].in = dims[l]; a.lay[l].out = dims[l + 1]; a.lay[l].act = 0;
    a.lay[l].w_off = off; off += dims[l] * dims[l + 1]; a.lay[l].b_off = off; off += dims[l + 1];
  }
  const int N = off;
  std::vector<double> h((size_t)N * (M + 1) + 12 * B);
  uint64_t s = 1;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 0.3; }
  double *d, *dZ, *dlp; int64_t* dn; long long* dst;
  hipMalloc(&d, h.size() * 8); hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&dZ, (size_t)M * itr * nch * 8); hipMalloc(&dlp, (size_t)itr * nch * 8); hipMalloc(&dn, nch * 8); hipMalloc(&dst, 64);
  a.swa = d; a.P = d + N; a.ldP = N; a.X = d + (size_t)N * (M + 1); a.Y = a.X + 10 * B;
  a.Z_out = dZ; a.lp_out = dlp; a.nacc_out = dn; a.itr = itr; a.seed = 1; a.sigma_z = 0.1; a.c0 = -183.8; a.sigma2 = 1.0;
  a.N = N; a.M = M; a.B = B; a.L = 3; a.chain_id0 = 0; a.slot_feats = 32; a.fuse_slots = 1; a.dbg_stamps = dst;
  const size_t lds = chain_loop_plan(a, 160 * 1024 - 256);
  printf("N %d, LDS %zu bytes, P in LDS %d\n", N, lds, a.p_in_lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    launch_chain_loop(0, a, nch, lds);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long st[8]; hipMemcpy(st, dst, 64, hipMemcpyDeviceToHost);
    printf("%lld transitions x %d chains: %.3f ms = %.2f us per transition | s_memtime ticks per transition (100 MHz?): propose %.0f  K4 %.0f  layers %.0f  fused+head %.0f  tail %.0f  final %.0f  accept %.0f\n",
           (long long)itr, nch, ms, ms * 1e3 / itr, (double)st[0] / itr, (double)st[1] / itr, (double)st[2] / itr, (double)st[3] / itr, (double)st[4] / itr, (double)st[5] / itr, (double)st[6] / itr);
  }
  return 0;
}
