// The shape-specialised fused forward (csrc/chain_spec.inc) against the generic one (kernels_chain_grid.hip) on the model of
// docs/src/nn_example.md (2-200-50-50-50-1, B = 1000): same bits of yhat, time per launch of `nchains` stacked chains.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DSI_SPEC_NB=<1|2|4>
//        -I subspaceinference.jl_amd/csrc tools/chain_spec_bench.hip -o tools/bin/chain_spec_bench_nb<NB>
// Run:   chain_spec_bench [nchains]
#include "../subspaceinference.jl_amd/csrc/kernels_chain_grid.hip"
#define SI_SPEC_L 5
#define SI_SPEC_DIMS 2, 200, 50, 50, 50, 1
#define SI_SPEC_ACTS 1, 1, 1, 1, 0
#define SI_SPEC_WOFF 0, 600, 10650, 13200, 15750
#define SI_SPEC_BOFF 400, 10600, 13150, 15700, 15800
#ifndef SI_SPEC_NB
#define SI_SPEC_NB 2
#endif
#define SI_SPEC_SF 32
#define SI_SPEC_M 20
#ifndef SI_SPEC_PREG
#define SI_SPEC_PREG 1
#endif
#include "../subspaceinference.jl_amd/csrc/chain_spec_args.h"
#include "../subspaceinference.jl_amd/csrc/chain_spec.inc"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
namespace si {
int32_t fail(Ctx*, int32_t c, const std::string&) { return c; }
ProfScope::ProfScope(Ctx*, int, double, double) {}
ProfScope::~ProfScope() {}
}
using namespace si;
int main(int argc, char** argv) {
  const int dims[6] = {2, 200, 50, 50, 50, 1}, acts[5] = {1, 1, 1, 1, 0}, B = 1000, L = 5;
  const int nch = argc > 1 ? atoi(argv[1]) : 512;
  const long long itr = argc > 2 ? atoll(argv[2]) : 0;   // > 0: also the persistent loop, `lch` chains
  const int lch = argc > 3 ? atoi(argv[3]) : 1;
  const int M = SI_SPEC_M;
  constexpr int NB = SI_SPEC_NB;
  si_layer lay[8] = {};
  int off = 0;
  for (int l = 0; l < L; ++l) {
    lay[l].kind = 0; lay[l].in = dims[l]; lay[l].out = dims[l + 1]; lay[l].act = acts[l];
    lay[l].w_off = off; off += dims[l] * dims[l + 1]; lay[l].b_off = off; off += dims[l + 1];
    if (lay[l].w_off != sispec::WOFF[l] || lay[l].b_off != sispec::BOFF[l]) { printf("offset table mismatch at layer %d: %lld %lld\n", l, (long long)lay[l].w_off, (long long)lay[l].b_off); return 1; }
  }
  const int N = off;
  const int64_t ldw = (N + 63) / 64 * 64;
  std::vector<double> h((size_t)ldw * std::max(nch, lch) + 3 * B + (size_t)ldw * (M + 1));
  uint64_t s = 1;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 0.3; }
  double *d, *dY0, *dY1;
  hipMalloc(&d, h.size() * 8); hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&dY0, (size_t)B * nch * 8); hipMalloc(&dY1, (size_t)B * nch * 8);
  hipMemset(dY0, 0xff, (size_t)B * nch * 8); hipMemset(dY1, 0x7f, (size_t)B * nch * 8);
  const double* X = d; double* W = d + 3 * B;
  std::vector<CgTileD> prog;
  int pstart[5], pcount[5], pchunks[5];
  chain_fused_program(lay, L, true, prog, pstart, pcount, pchunks);
  CgTileD* dprog; hipMalloc(&dprog, prog.size() * sizeof(CgTileD)); hipMemcpy(dprog, prog.data(), prog.size() * sizeof(CgTileD), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  ChainFusedPlan p{};
  const size_t lds = chain_fused_plan(p, lay, L, B, NB, true, 32, 2);
  p.prog = dprog;
  for (int i = 0; i < 5; ++i) { p.prog_start[i] = pstart[i]; p.prog_count[i] = pcount[i]; p.prog_chunks[i] = pchunks[i]; }
  const size_t lds_spec = (size_t)sispec::LDS_DOUBLES * 8;
  const dim3 grid((B + 16 * NB - 1) / (16 * NB), nch);
  if (lds_spec > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(si_spec_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_spec);
  printf("N %d, %d chains, NB %d: generic LDS %zu B, specialised LDS %zu B, grid %u x %u\n", N, nch, NB, lds, lds_spec, grid.x, grid.y);
  auto run_generic = [&]() { launch_chain_fused(0, p, NB, false, lds, W, ldw, X, dY0, B, nch); };
  auto run_spec = [&]() { hipLaunchKernelGGL(si_spec_fused_kernel, grid, dim3(256), lds_spec, 0, W, (long long)ldw, X, dY1, (long long)B, B); };
  double* dY2; hipMalloc(&dY2, (size_t)B * nch * 8); hipMemset(dY2, 0x3f, (size_t)B * nch * 8);
  const size_t lds_stack = (size_t)sispec::WVEC * 8;
  const int splits = nch >= 256 ? 1 : std::min(8, (256 + nch - 1) / nch);
  hipFuncSetAttribute(reinterpret_cast<const void*>(si_spec_stack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stack);
  auto run_stack = [&]() { hipLaunchKernelGGL(si_spec_stack_kernel, dim3(splits, nch), dim3(512), lds_stack, 0, W, (long long)ldw, X, dY2, (long long)B, B); };
  run_generic(); run_spec(); run_stack();
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
  std::vector<double> y0((size_t)B * nch), y1((size_t)B * nch);
  hipMemcpy(y0.data(), dY0, y0.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(y1.data(), dY1, y1.size() * 8, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t i = 0; i < y0.size(); ++i) diff += memcmp(&y0[i], &y1[i], 8) != 0;
  printf("yhat: %zu of %zu values differ in their bits (y[0] = %.17g / %.17g)\n", diff, y0.size(), y0[0], y1[0]);
  {
    std::vector<double> y2((size_t)B * nch);
    hipMemcpy(y2.data(), dY2, y2.size() * 8, hipMemcpyDeviceToHost);
    size_t d2 = 0; double maxd = 0;
    for (size_t i = 0; i < y0.size(); ++i) { d2 += memcmp(&y0[i], &y2[i], 8) != 0; maxd = fmax(maxd, fabs(y0[i] - y2[i])); }
    printf("register-resident stacked kernel (%d splits x %d chains, LDS %zu B): %zu of %zu values differ in their bits, max |diff| %.3e (y[0] = %.17g)\n", splits, nch, lds_stack, d2, y0.size(), maxd, y2[0]);
    diff += d2;
  }
  for (int which = 0; which < 3; ++which)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) which == 2 ? run_stack() : which ? run_spec() : run_generic();
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("  %-11s %.1f us per launch = %.2f TFLOP/s (useful 2 N B per chain)\n", which == 2 ? "stacked-reg" : which ? "specialised" : "generic", ms * 100.0, 2.0 * N * B * nch / (ms * 1e-4) / 1e12);
    }
  if (itr > 0) {   // the persistent loop: generic against specialised, Z and lp bit for bit
    const double* swa = W + (size_t)ldw * std::max(nch, lch); const double* P = swa + ldw; const double* Y = X + 2 * B;
    double *dZ[2], *dlp[2], *dYl; long long* dn; unsigned* dsync;
    for (int k = 0; k < 2; ++k) { hipMalloc(&dZ[k], (size_t)M * itr * lch * 8); hipMalloc(&dlp[k], (size_t)itr * lch * 8); }
    hipMalloc(&dn, lch * 8); hipMalloc(&dsync, 128 * (8 * lch + 1)); hipMalloc(&dYl, (size_t)2 * B * lch * 8);
    const long long ldf = sispec::FO_TOTAL; double* dW4; hipMalloc(&dW4, (size_t)4 * ldf * lch * 8); hipMemset(dW4, 0, (size_t)4 * ldf * lch * 8);
    int* dperm; hipMalloc(&dperm, (size_t)N * 4); hipMemset(dperm, 0xff, (size_t)N * 4);
    hipLaunchKernelGGL(si_spec_perm_kernel, dim3((200 * 50 + 255) / 256), dim3(256), 0, 0, dperm);
    { std::vector<int> hp(N); hipMemcpy(hp.data(), dperm, (size_t)N * 4, hipMemcpyDeviceToHost); std::vector<char> seen(ldf, 0); int bad = 0; for (int r = 0; r < N; ++r) { if (hp[r] < 0 || hp[r] >= ldf || seen[hp[r]]) ++bad; else seen[hp[r]] = 1; } printf("fragment order: %lld doubles for N = %d, %d bad entries of the permutation\n", ldf, N, bad); if (bad) return 1; }
    ChainGridArgs a{};
    const int G = (B + 16 * NB - 1) / (16 * NB);
    const size_t lf = chain_fused_plan(a.p, lay, L, B, NB, true, 32, 2);
    a.p.prog = dprog;
    for (int i = 0; i < 5; ++i) { a.p.prog_start[i] = pstart[i]; a.p.prog_count[i] = pcount[i]; a.p.prog_chunks[i] = pchunks[i]; }
    a.M = M; a.nblocks = 4; a.G = G;
    const size_t ldsg = chain_grid_plan(a, lf);
    a.swa = swa; a.P = P; a.X = X; a.Y = Y; a.wbuf = W; a.w_stride = ldw; a.ybuf = dYl; a.y_stride = B;
    a.cnt = dsync; a.status = dsync + 32 * 8 * lch; a.Z_out = dZ[0]; a.lp_out = dlp[0]; a.nacc_out = (int64_t*)dn; a.ldP = ldw; a.itr = itr;
    a.seed = 1; a.sigma_z = 0.1; a.c0 = -918.9; a.sigma2 = 1.0; a.N = N; a.chain_id0 = 0;
    SiSpecGridArgs sa{};
    sa.swa = swa; sa.P = P; sa.X = X; sa.Y = Y; sa.wbuf = dW4; sa.w_stride = ldf; sa.perm = dperm; sa.ybuf = dYl; sa.y_stride = B; sa.cnt = dsync; sa.status = dsync + 32 * 8 * lch;
    sa.Z_out = dZ[1]; sa.lp_out = dlp[1]; sa.nacc_out = dn; sa.ldP = ldw; sa.itr = itr; sa.seed = 1; sa.sigma_z = 0.1; sa.c0 = -918.9; sa.sigma2 = 1.0;
    sa.N = N; sa.M = M; sa.G = G; sa.B = B; sa.chain_id0 = 0; sa.nblocks = 4;
    {   // the loop's LDS behind the specialised images (chain_grid_plan's layout)
      auto even = [](long long v) { return (v + 1) & ~1ll; };
      long long o = sispec::LDS_DOUBLES; const long long d = (long long)sispec::OUTL * B;
      sa.y_in_lds = d <= 4096; sa.o_y = (int)o; o += even(sa.y_in_lds ? d : 0); sa.o_blk = (int)o; o += even(sa.nblocks);
      sa.o_z = (int)o; o += even(5ll * M); sa.o_red = (int)o; o += 24; sa.o_flag = (int)o; o += 2;
      const size_t ldss = std::max((size_t)o * 8, (size_t)(81 * 1024));
      hipFuncSetAttribute(reinterpret_cast<const void*>(si_spec_grid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldss);
      printf("loop: %d chains x G %d workgroups, NB %d, %lld transitions; LDS generic %zu B, specialised %zu B; rows of P per workgroup %d (PREG %d)\n", lch, G, NB, itr, ldsg, ldss,
             ((N + G - 1) / G + 15) & ~15, (int)SI_SPEC_PREG);
      if (G * lch > 256 || (SI_SPEC_PREG && (((N + G - 1) / G + 15) & ~15) > 256)) { printf("does not apply\n"); return 1; }
      for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(dsync, 0, 128 * (8 * lch + 1));
          const auto wt0 = std::chrono::steady_clock::now();
          hipEventRecord(e0, 0);
          static hipStream_t lst = nullptr;
          if (!lst && getenv("SI_HARNESS_STREAM")) hipStreamCreateWithFlags(&lst, hipStreamNonBlocking);   // (the library launches on a stream of its own)
          if (which) hipLaunchKernelGGL(si_spec_grid_kernel, dim3(G * lch), dim3(256), ldss, lst, sa); else launch_chain_grid(0, a, NB, lch, ldsg);
          if (lst) hipStreamSynchronize(lst);
          hipEventRecord(e1, 0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          unsigned stt; hipMemcpy(&stt, sa.status, 4, hipMemcpyDeviceToHost);
          const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wt0).count() * 1e3;
          printf("  %-11s %.3f ms = %.2f us per transition (host wall %.3f ms), status %u\n", which ? "specialised" : "generic", ms, ms * 1e3 / itr, wall, stt);
        }
    }
    std::vector<double> z0((size_t)M * itr * lch), z1(z0.size()), l0((size_t)itr * lch), l1(l0.size());
    hipMemcpy(z0.data(), dZ[0], z0.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(z1.data(), dZ[1], z1.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(l0.data(), dlp[0], l0.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(l1.data(), dlp[1], l1.size() * 8, hipMemcpyDeviceToHost);
    size_t dz = 0, dl = 0;
    for (size_t i = 0; i < z0.size(); ++i) dz += memcmp(&z0[i], &z1[i], 8) != 0;
    for (size_t i = 0; i < l0.size(); ++i) dl += memcmp(&l0[i], &l1[i], 8) != 0;
#ifdef SI_SPEC_STAMPS
    {
      unsigned long long st[24];
      hipMemcpyFromSymbol(st, HIP_SYMBOL(sispec::stamp_sum), sizeof(st));
      const char* nm[24] = {"", "preload issued", "L0", "sync", "L1", "sync", "L2", "sync", "L3", "sync", "L4", "sync", "head", "sync", "tail of forward", "candidates + K4", "arrive", "wait", "SSE loads + tree", "final + accept", "", "", "", ""};
      double tot = 0;
      printf("  cycles per transition, workgroup 0 wave 0:");
      for (int i = 1; i < 20; ++i) if (st[i]) { printf(" [%s] %.0f", nm[i], (double)st[i] / itr); tot += (double)st[i] / itr; }
      printf("  total %.0f\n", tot);
    }
#endif
    printf("loop: %zu of %zu z values and %zu of %zu lp values differ in their bits (lp[last] = %.17g / %.17g)\n", dz, z0.size(), dl, l0.size(), l0.back(), l1.back());
    diff += dz + dl;
  }
#ifdef SI_SPEC_STAMPS
  {
    unsigned long long z[24] = {0}, st[24];
    hipMemcpyToSymbol(HIP_SYMBOL(sispec::stamp_sum), z, sizeof(z));
    run_spec(); hipDeviceSynchronize();
    hipMemcpyFromSymbol(st, HIP_SYMBOL(sispec::stamp_sum), sizeof(st));
    const double per = (double)grid.x * grid.y;
    printf("  s_memtime ticks (10 ns) per workgroup: [0] X staged");
    const char* nm[16] = {"", "", "L0", "sync", "L1", "sync", "L2", "sync", "L3", "sync", "L4", "sync", "head", "sync", "tail", ""};
    double tot = 0;
    for (int i = 0; i < 15; ++i) { if (i) printf(" [%s] %.1f", nm[i], st[i] / per); else printf(" %.1f", st[0] / per); tot += st[i] / per; }
    printf("  total %.1f\n", tot);
  }
#endif
  return diff != 0;
}
