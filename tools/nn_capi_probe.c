/* the nn_example model through the C ABI WITHOUT Python: si_create, si_infer_setup, si_sample_rwmh -- wall clock per transition
 * (is the 9 % between a plain Python process and the C harness the interpreter's or the library's?)
 * build: gcc -O2 -I include tools/nn_capi_probe.c -o tools/bin/nn_capi_probe -ldl -lm */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "subspace_hip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char** argv) {
  void* h = dlopen(argc > 1 ? argv[1] : "subspaceinference.jl_amd/libsubspace_hip.so", RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  int32_t (*create)(si_ctx**, int32_t) = dlsym(h, "si_create");
  int32_t (*setup)(si_ctx*, const si_layer*, int32_t, int64_t, int32_t, const double*, const double*, const double*, const double*, int32_t, int32_t, int64_t, double, int32_t) = dlsym(h, "si_infer_setup");
  int32_t (*sample)(si_ctx*, int64_t, double, uint64_t, int32_t, int32_t, double*, double*, double*) = dlsym(h, "si_sample_rwmh");
  const char* (*lasterr)(si_ctx*) = dlsym(h, "si_last_error");
  const int dims[6] = {2, 200, 50, 50, 50, 1}, acts[5] = {1, 1, 1, 1, 0};
  const int L = 5, B = 1000, M = 20;
  si_layer lay[5];
  int64_t off = 0;
  for (int l = 0; l < L; ++l) {
    si_layer z = {0};
    lay[l] = z;
    lay[l].in = dims[l]; lay[l].out = dims[l + 1]; lay[l].act = acts[l]; lay[l].w_off = off; off += (int64_t)dims[l] * dims[l + 1];
    lay[l].b_off = off; off += dims[l + 1];
  }
  const int64_t N = off;
  double* w = malloc(N * 8), *P = malloc(N * M * 8), *X = malloc(2 * B * 8), *Y = malloc(B * 8);
  uint64_t s = 1;
#define RND() (s = s * 6364136223846793005ull + 1442695040888963407ull, ((double)(s >> 11) / 9007199254740992.0 - 0.5))
  for (int64_t i = 0; i < N; ++i) w[i] = 0.6 * RND();
  for (int64_t i = 0; i < N * M; ++i) P[i] = 0.1 * RND();
  for (int i = 0; i < 2 * B; ++i) X[i] = 2.0 * RND();
  for (int i = 0; i < B; ++i) Y[i] = 2.0 * RND();
  si_ctx* ctx = NULL;
  if (create(&ctx, 0) != 0) { printf("si_create failed: %s\n", lasterr(NULL)); return 1; }
  if (setup(ctx, lay, L, N, M, w, P, X, Y, 2, 1, B, 1.0, 1) != 0) { printf("setup: %s\n", lasterr(ctx)); return 1; }
  const int64_t itr = 20000;
  double* Z = malloc(M * itr * 8), *lp = malloc(itr * 8), acc;
  sample(ctx, 20, 0.1, 1, 0, 1, Z, lp, &acc);
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = now();
    if (sample(ctx, itr, 0.1, 1, 0, 1, Z, lp, &acc) != 0) { printf("sample: %s\n", lasterr(ctx)); return 1; }
    printf("C ABI without Python: %.2f us per transition (lp[last] %.6f)\n", (now() - t0) / itr * 1e6, lp[itr - 1]);
  }
  return 0;
}
