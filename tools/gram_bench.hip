// Development harness for K2 (Gram) and K3 (projection) at construct shapes.  Not shipped.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gram_bench.hip subspaceinference.jl_amd/csrc/build/kernels_gram_wave{0,1,2}.o -o tools/bin/gram_bench
#include "../subspaceinference.jl_amd/csrc/kernels_gram.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
namespace si {
int32_t fail(Ctx*, int32_t c, const std::string&) { return c; }
ProfScope::ProfScope(Ctx*, int, double, double) {}
ProfScope::~ProfScope() {}
// the wide-subspace projection lives in kernels_bwd.hip; this harness only exercises M <= 32
void launch_project_mfma(hipStream_t, const double*, int64_t, int64_t, int64_t, const double*, int32_t, int32_t, double*, int64_t) {}
bool launch_project_stream(hipStream_t, const double*, int64_t, int64_t, int64_t, const double*, int32_t, int32_t, double*, int64_t, int) { return false; }
bool launch_project_stream_f32(hipStream_t, const float*, int64_t, int64_t, int64_t, const double*, int32_t, int32_t, double*, int64_t, int) { return false; }
}
using namespace si;
#ifdef SI_GRAM_TS
namespace si { void gram_spec_dump_ts(unsigned long long* host); }
#endif
__global__ void fill(double* a, size_t n, unsigned long long seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long s = (i + 1) * 6364136223846793005ull + seed; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
    a[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
  }
}
// reference: one workgroup per (i, j), plain fp64 dot product with a fixed-order tree (development check of the MFMA kernels)
__global__ __launch_bounds__(256) void gram_naive(const double* A, int64_t ldA, int64_t N, int K, double* G) {
  const int i = blockIdx.x, j = blockIdx.y;
  if (i > j) return;
  __shared__ double red[256];
  double acc = 0.0;
  for (int64_t r = threadIdx.x; r < N; r += 256) acc += A[r + (int64_t)i * ldA] * A[r + (int64_t)j * ldA];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) G[i + (int64_t)K * j] = G[j + (int64_t)K * i] = red[0];
}
int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 1047361;
  const int K = argc > 2 ? atoi(argv[2]) : 100, M = argc > 3 ? atoi(argv[3]) : 20;
  const int64_t ldA = pad_ld(N);
  double *A, *G, *Gp, *V, *P;
  hipMalloc(&A, (size_t)ldA * K * 8); hipMemset(A, 0, (size_t)ldA * K * 8);
  for (int k = 0; k < K; ++k) hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, A + (size_t)k * ldA, (size_t)N, 1000ull + k);
  hipMalloc(&G, (size_t)K * K * 8);
  const int NCU = getenv("SI_BENCH_CUS") ? atoi(getenv("SI_BENCH_CUS")) : 256;   // development: fewer workgroups than CUs
  const size_t need = launch_gram(0, A, ldA, N, K, nullptr, nullptr, NCU, nullptr);
  hipMalloc(&Gp, need); hipMemset(Gp, 0, need);
  const int Mpad = project_mpad(M);
  hipMalloc(&V, (size_t)K * Mpad * 8); hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, V, (size_t)K * Mpad, 7ull);
  hipMalloc(&P, (size_t)ldA * M * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int REPS = getenv("SI_BENCH_REPS") ? atoi(getenv("SI_BENCH_REPS")) : 4;   // development: long runs for rocm-smi sampling
  for (int rep = 0; rep < REPS; ++rep) {
    hipEventRecord(e0, 0);
    for (int it = 0; it < 5; ++it) launch_gram(0, A, ldA, N, K, Gp, G, NCU, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    hipEventRecord(e0, 0);
    launch_project(0, A, ldA, N, K, V, M, Mpad, P, ldA, 256);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms2; hipEventElapsedTime(&ms2, e0, e1);
    if (rep < 4 || rep == REPS - 1) printf("N=%lld K=%d M=%d: gram+reduce %.3f ms (%.2f TFLOP/s useful, %.2f TB/s of A), project %.3f ms (%.2f TB/s)\n", (long long)N, K, M, ms,
           (double)N * K * (K + 1) / (ms * 1e-3) / 1e12, (double)N * K * 8 / (ms * 1e-3) / 1e12, ms2, (double)N * (K + M) * 8 / (ms2 * 1e-3) / 1e12);
  }
#ifdef SI_GRAM_TS
  {  // wave-specialised kernel only (SI_GRAM_SPEC=1): per-slab barrier wait and slab period of 8 sampled workgroups, 10 ns ticks
    std::vector<unsigned long long> ts(8 * 1024);
    gram_spec_dump_ts(ts.data());
    for (int b = 0; b < 8; b += 3) {
      printf("block %d: [period wait] in 10 ns ticks per slab:", 32 * b);
      for (int it = 1; it < 140 && ts[b * 1024 + 2 * it]; ++it)
        printf(" %llu/%llu", ts[b * 1024 + 2 * it + 1] - ts[b * 1024 + 2 * it - 1], ts[b * 1024 + 2 * it + 1] - ts[b * 1024 + 2 * it]);
      printf("\n");
    }
  }
#endif
  std::vector<double> g((size_t)K * K); hipMemcpy(g.data(), G, g.size() * 8, hipMemcpyDeviceToHost);
  {
    double* Gn; hipMalloc(&Gn, (size_t)K * K * 8);
    hipLaunchKernelGGL(gram_naive, dim3(K, K), dim3(256), 0, 0, A, ldA, N, K, Gn);
    std::vector<double> gn((size_t)K * K); hipMemcpy(gn.data(), Gn, gn.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0.0; int wi = 0, wj = 0; bool sym = true;
    for (int j = 0; j < K; ++j) for (int i = 0; i < K; ++i) {
      const double sc = sqrt(gn[(size_t)i * K + i] * gn[(size_t)j * K + j]);
      const double e = fabs(g[i + (size_t)K * j] - gn[i + (size_t)K * j]) / sc;
      if (!(e <= worst)) { worst = e; wi = i; wj = j; }
      if (g[i + (size_t)K * j] != g[j + (size_t)K * i]) sym = false;
    }
    printf("CHECK K=%d: max |G - G_naive| / sqrt(G_ii G_jj) = %.3e at (%d, %d), symmetric %d -> %s\n", K, worst, wi, wj, (int)sym,
           (worst < 1e-12 && sym) ? "OK" : "FAIL");
  }
  { unsigned long long x = 0; for (size_t i = 0; i < g.size(); ++i) { unsigned long long u; memcpy(&u, &g[i], 8); x = (x * 1099511628211ull) ^ u; } printf("G bits hash %016llx\n", x); }
  printf("G[0,0]=%.6f G[1,0]=%.6f G[K-1,K-1]=%.6f (expect ~N/12=%.1f on the diagonal)\n", g[0], g[1], g[(size_t)K * K - 1], N / 12.0);
  return 0;
}
