// Where the fused Dense-chain forward and the persistent grid loop (kernels_chain_grid.hip) spend their cycles, on the model of
// docs/src/nn_example.md (2-200-50-50-50-1, B = 1000, M = 20): s_memtime stamps of workgroup 0.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DSI_CG_STAMPS
//        -I subspaceinference.jl_amd/csrc tools/chain_grid_bench.hip -o tools/bin/chain_grid_bench
// Run:   chain_grid_bench [nchains of the stacked launch] [NB] [itr of the loop] [chains of the loop]
#include "../subspaceinference.jl_amd/csrc/kernels_chain_grid.hip"
#include <cstdio>
#include <vector>
namespace si {
int32_t fail(Ctx*, int32_t c, const std::string&) { return c; }
ProfScope::ProfScope(Ctx*, int, double, double) {}
ProfScope::~ProfScope() {}
}
using namespace si;
static void stamps(const char* what, double per) {
#ifndef SI_CG_STAMPS
  return;
#else
  long long st[32];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(cg_stamp_sum), sizeof(st));
  printf("  %s, s_memtime ticks (100 MHz) per unit:", what);
  for (int i = 0; i < 18; ++i) printf(" [%d] %.1f", i, (double)st[i] / per);
  printf("\n");
  long long z[32] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(cg_stamp_sum), z, sizeof(z));
#endif
}
int main(int argc, char** argv) {
  const int dims[6] = {2, 200, 50, 50, 50, 1}, acts[5] = {1, 1, 1, 1, 0}, B = 1000, M = 20, L = 5;
  const int nch = argc > 1 ? atoi(argv[1]) : 512;
  const int NB = argc > 2 ? atoi(argv[2]) : 2;
  const int64_t itr = argc > 3 ? atoll(argv[3]) : 2000;
  const int lch = argc > 4 ? atoi(argv[4]) : 1;
  const bool wave_tiles = argc > 5 ? atoi(argv[5]) != 0 : true;
#ifdef SI_CG_KNOB
  printf("KNOB %d (compile-time; 1 no W loads, 2 no H reads, 4 no MFMAs, 8 no image stores)\n", SI_CG_KNOB);
#endif
  si_layer lay[8] = {};
  int off = 0;
  for (int l = 0; l < L; ++l) {
    lay[l].kind = 0; lay[l].in = dims[l]; lay[l].out = dims[l + 1]; lay[l].act = acts[l];
    lay[l].w_off = off; off += dims[l] * dims[l + 1]; lay[l].b_off = off; off += dims[l + 1];
  }
  const int N = off;
  const int64_t ldw = (N + 63) / 64 * 64;
  std::vector<double> h((size_t)ldw * (M + 1) + 3 * B + (size_t)ldw * std::max(nch, lch));   // (the weight slots serve both legs)
  uint64_t s = 1;
  for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 0.3; }
  double *d, *dY, *dZ, *dlp; int64_t* dn; unsigned* dsync;
  hipMalloc(&d, h.size() * 8); hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipMalloc(&dY, (size_t)B * std::max(nch, lch) * 8);
  hipMalloc(&dZ, (size_t)M * itr * lch * 8); hipMalloc(&dlp, (size_t)itr * lch * 8); hipMalloc(&dn, lch * 8); hipMalloc(&dsync, 128 * (lch + 1));
  const double* swa = d; const double* P = d + ldw; const double* X = d + (size_t)ldw * (M + 1); const double* Y = X + 2 * B;
  double* W = d + (size_t)ldw * (M + 1) + 3 * B;
  std::vector<CgTileD> prog;
  int pstart[5], pcount[5], pchunks[5];
  chain_fused_program(lay, L, true, prog, pstart, pcount, pchunks);
  CgTileD* dprog; hipMalloc(&dprog, prog.size() * sizeof(CgTileD)); hipMemcpy(dprog, prog.data(), prog.size() * sizeof(CgTileD), hipMemcpyHostToDevice);
  printf("tile program: %zu tiles; one wave per batch tile: %d tiles, %d chunks of 16 k\n", prog.size(), pcount[0], pchunks[0]);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {   // the stacked one-launch density
    ChainFusedPlan p{};
    const size_t lds = chain_fused_plan(p, lay, L, B, NB, true, 32, 2);
    p.prog = dprog;
    for (int i = 0; i < 5; ++i) { p.prog_start[i] = pstart[i]; p.prog_count[i] = pcount[i]; p.prog_chunks[i] = pchunks[i]; }
    printf("fused forward: N %d, %d chains, NB %d, %s per tile, LDS %zu bytes per tile, %d tiles\n", N, nch, NB, wave_tiles ? "a wave" : "a workgroup", lds, (B + 16 * NB - 1) / (16 * NB) * nch);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) launch_chain_fused(0, p, NB, wave_tiles, lds, W, ldw, X, dY, B, nch);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("  %.1f us per launch = %.2f TFLOP/s (useful 2 N B per chain)\n", ms * 100.0, 2.0 * N * B * nch / (ms * 1e-4) / 1e12);
      stamps("workgroup (0, 0)", 10.0);
    }
  }
  {   // the persistent loop
    ChainGridArgs a{};
    int nb = 0; size_t lds = 0;
    for (int cand : {1, 2, 4}) {
      const int G = (B + 16 * cand - 1) / (16 * cand);
      if (G * lch > 256) continue;
      const size_t lf = chain_fused_plan(a.p, lay, L, B, cand, true, 32, 2);
      a.p.prog = dprog;
      for (int i = 0; i < 5; ++i) { a.p.prog_start[i] = pstart[i]; a.p.prog_count[i] = pcount[i]; a.p.prog_chunks[i] = pchunks[i]; }
      a.M = M; a.nblocks = 4; a.G = G;
      lds = chain_grid_plan(a, lf);
      if (lds) { nb = cand; break; }
    }
    a.swa = swa; a.P = P; a.X = X; a.Y = Y; a.wbuf = W; a.w_stride = ldw; a.ybuf = dY; a.y_stride = B;
    a.cnt = dsync; a.status = dsync + 32 * lch; a.Z_out = dZ; a.lp_out = dlp; a.nacc_out = dn; a.ldP = ldw; a.itr = itr;
    a.seed = 1; a.sigma_z = 0.1; a.c0 = -918.9; a.sigma2 = 1.0; a.N = N; a.chain_id0 = 0;
    printf("grid loop: %d chains, NB %d, G %d, LDS %zu bytes\n", lch, nb, a.G, lds);
    for (int rep = 0; rep < 3 && nb; ++rep) {
      hipMemset(dsync, 0, 128 * (lch + 1));
      hipEventRecord(e0, 0);
      launch_chain_grid(0, a, nb, lch, lds);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned st; hipMemcpy(&st, a.status, 4, hipMemcpyDeviceToHost);
      printf("  %lld transitions: %.3f ms = %.2f us per transition, status %u\n", (long long)itr, ms, ms * 1e3 / itr, st);
      stamps("workgroup 0: [0] propose [1] K4 [2] arrive [3] wait A [8..] layers [16] head [4] rest of forward [5] barrier B [6] tail + accept", (double)itr);
    }
  }
  return 0;
}
