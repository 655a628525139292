/* CPU restatement in C + OpenMP of the DENSITY of SubspaceInference.jl's hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library (through
 * oracle/c_port.py).  The product (subspaceinference.jl_amd) never does.  PARITY UNPINNED, like the NumPy restatement
 * it mirrors (oracle/subspace_oracle.py; tests/test_oracle.py holds the two against each other): the reference ships
 * no fixtures and Julia is absent.
 *
 * What it restates (paths relative to /root/reference):
 *   so_logdensity     src/space_inference.jl:90-95   new_W = W_swa + P*z; new_model = model_re(in_model, new_W);
 *                                                     logpdf(MvNormal(vec(new_model(in_data)), sigma_m), vec(out_data))
 *                                                     -- likelihood only, the prior line after `return` is dead code (Q4)
 *   dense layers      src/libs.jl:55-57 (re(W): column-major slices) + Flux 0.11.2 Dense  sigma.(W*x .+ b)
 *
 * Why it exists: SURVEY 8(d) asks for the CPU baseline "timed single-core and all-core" on a compiled port; the NumPy
 * port spends its time in single-threaded elementwise passes over 768 MB temporaries.  Here the forward pass is ONE
 * blocked fp64 GEMM per layer with the bias + activation in its epilogue (what Julia's BLAS + fused broadcast do at
 * best), parallel over observation blocks, AVX-512 / AVX2+FMA micro-kernels chosen at run time.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int32_t in, out, act; /* act: 0 identity, 1 relu, 2 tanh, 3 sigmoid, 4 leakyrelu, 5 elu, 6 softplus, 7 selu (oracle/subspace_oracle.py ACT_*) */
  int64_t w_off, b_off;
} so_layer;

static inline double act_apply(double a, int act) {
  switch (act) {
    case 1: return a > 0.0 ? a : 0.0;
    case 2: return tanh(a);
    case 3: return 1.0 / (1.0 + exp(-a));
    case 4: return a > 0.01 * a ? a : 0.01 * a;
    case 5: return a >= 0.0 ? a : exp(a) - 1.0;   /* NNlib: alpha * (exp(x) - one(x)) */
    case 6: return a > 0.0 ? a + log1p(exp(-a)) : log1p(exp(a));
    case 7: return 1.0507009873554805 * (a > 0.0 ? a : 1.6732632423543772 * (exp(a) - 1.0));
    default: return a;
  }
}

/* ---- micro-kernels: C[MR x NR] += Wp[MR x in] (packed, k-major) * H[in x NR] (column j at H + j*ldh) ------------- */
typedef double v8d __attribute__((vector_size(64), aligned(8)));
typedef double v4d __attribute__((vector_size(32), aligned(8)));

#define MR512 16
#define NR512 12
__attribute__((target("avx512f"))) static void micro_avx512(const double* wp, const double* h, int64_t ldh, int in,
                                                            const double* bias, int act, double* c, int64_t ldc) {
  v8d acc0[NR512], acc1[NR512];
  for (int j = 0; j < NR512; ++j) {
    acc0[j] = (v8d){0, 0, 0, 0, 0, 0, 0, 0};
    acc1[j] = acc0[j];
  }
  for (int k = 0; k < in; ++k) {
    const v8d a0 = *(const v8d*)(wp + (size_t)k * MR512);
    const v8d a1 = *(const v8d*)(wp + (size_t)k * MR512 + 8);
#pragma GCC unroll 12
    for (int j = 0; j < NR512; ++j) {
      const double b = h[(size_t)j * ldh + k];
      const v8d bb = {b, b, b, b, b, b, b, b};
      acc0[j] += a0 * bb;
      acc1[j] += a1 * bb;
    }
  }
  for (int j = 0; j < NR512; ++j)
    for (int i = 0; i < 8; ++i) {
      c[(size_t)j * ldc + i] = act_apply(acc0[j][i] + bias[i], act);
      c[(size_t)j * ldc + 8 + i] = act_apply(acc1[j][i] + bias[8 + i], act);
    }
}

#define MR256 8
#define NR256 6
__attribute__((target("avx2,fma"))) static void micro_avx2(const double* wp, const double* h, int64_t ldh, int in,
                                                           const double* bias, int act, double* c, int64_t ldc) {
  v4d acc0[NR256], acc1[NR256];
  for (int j = 0; j < NR256; ++j) {
    acc0[j] = (v4d){0, 0, 0, 0};
    acc1[j] = acc0[j];
  }
  for (int k = 0; k < in; ++k) {
    const v4d a0 = *(const v4d*)(wp + (size_t)k * MR256);
    const v4d a1 = *(const v4d*)(wp + (size_t)k * MR256 + 4);
#pragma GCC unroll 6
    for (int j = 0; j < NR256; ++j) {
      const double b = h[(size_t)j * ldh + k];
      const v4d bb = {b, b, b, b};
      acc0[j] += a0 * bb;
      acc1[j] += a1 * bb;
    }
  }
  for (int j = 0; j < NR256; ++j)
    for (int i = 0; i < 4; ++i) {
      c[(size_t)j * ldc + i] = act_apply(acc0[j][i] + bias[i], act);
      c[(size_t)j * ldc + 4 + i] = act_apply(acc1[j][i] + bias[4 + i], act);
    }
}

/* portable edge / fallback: rows [i0, i1) x columns [j0, j1) straight from the unpacked W */
static void block_scalar(const double* W, const double* bias, const double* H, double* C, int out, int in, int i0, int i1,
                         int64_t j0, int64_t j1, int act) {
  for (int64_t j = j0; j < j1; ++j) {
    const double* hj = H + (size_t)j * in;
    for (int i = i0; i < i1; ++i) {
      double s = 0.0;
      for (int k = 0; k < in; ++k) s += W[(size_t)k * out + i] * hj[k];
      C[(size_t)j * out + i] = act_apply(s + bias[i], act);
    }
  }
}

static int g_isa = -1; /* 2 = avx512f, 1 = avx2+fma, 0 = scalar */
static int isa(void) {
  if (g_isa < 0) {
    __builtin_cpu_init();
    g_isa = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
  }
  return g_isa;
}
int so_isa(void) { return isa(); }
int so_max_threads(void) { return omp_get_num_procs(); }

/* C (out x B) = act.(W (out x in, column-major) * H (in x B) .+ b) */
static void dense_forward(const double* W, const double* bias, const double* H, double* C, int out, int in, int64_t B, int act,
                          int threads) {
  const int kind = isa();
  const int MR = kind == 2 ? MR512 : MR256, NR = kind == 2 ? NR512 : NR256;
  const int64_t NB = (int64_t)NR * 4; /* observation block per task: H block (in x NB) stays in L2 */
  const int64_t nblk = (B + NB - 1) / NB;
  if (kind == 0 || out < MR) {
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t t = 0; t < nblk; ++t) {
      const int64_t j0 = t * NB, j1 = j0 + NB < B ? j0 + NB : B;
      block_scalar(W, bias, H, C, out, in, 0, out, j0, j1, act);
    }
    return;
  }
  const int full_rows = out / MR * MR;
#pragma omp parallel num_threads(threads)
  {
    double* wp = (double*)aligned_alloc(64, (size_t)MR * (size_t)in * sizeof(double) + 64);
#pragma omp for schedule(dynamic, 4)
    for (int64_t t = 0; t < nblk; ++t) {
      const int64_t j0 = t * NB, j1 = j0 + NB < B ? j0 + NB : B;
      const int64_t jfull = j0 + (j1 - j0) / NR * NR;
      for (int i0 = 0; i0 < full_rows; i0 += MR) {
        for (int k = 0; k < in; ++k) memcpy(wp + (size_t)k * MR, W + (size_t)k * out + i0, (size_t)MR * sizeof(double));
        for (int64_t j = j0; j < jfull; j += NR) {
          if (kind == 2)
            micro_avx512(wp, H + (size_t)j * in, in, in, bias + i0, act, C + (size_t)j * out + i0, out);
          else
            micro_avx2(wp, H + (size_t)j * in, in, in, bias + i0, act, C + (size_t)j * out + i0, out);
        }
      }
      if (jfull < j1) block_scalar(W, bias, H, C, out, in, 0, full_rows, jfull, j1, act);
      if (full_rows < out) block_scalar(W, bias, H, C, out, in, full_rows, out, j0, j1, act);
    }
    free(wp);
  }
}

/* src/space_inference.jl:91  new_W = W_swa + P*z  (P is N x M column-major) */
void so_reconstruct(const double* w_swa, const double* P, int64_t N, int32_t M, const double* z, double* w, int threads) {
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int64_t i = 0; i < N; ++i) {
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += P[(size_t)m * N + i] * z[m];
    w[i] = w_swa[i] + s;
  }
}

/* workspace kept between calls (a fresh 768 MB malloc per evaluation costs more in page faults than the GEMM of a
 * small layer; Julia's GC-managed temporaries would pay that too -- the baseline is meant to flatter the CPU) */
static double* g_ws[4] = {NULL, NULL, NULL, NULL};
static size_t g_ws_cap[4] = {0, 0, 0, 0};
static double* ws_get(int slot, size_t elems) {
  if (g_ws_cap[slot] < elems) {
    free(g_ws[slot]);
    g_ws[slot] = (double*)malloc(elems * sizeof(double));
    g_ws_cap[slot] = g_ws[slot] ? elems : 0;
  }
  return g_ws[slot];
}
void so_release(void) {
  for (int i = 0; i < 4; ++i) {
    free(g_ws[i]);
    g_ws[i] = NULL;
    g_ws_cap[i] = 0;
  }
}

/* forward of the whole chain on the flat weight vector; yhat_out is out_dim x B.  Returns 0, or -1 on allocation failure. */
int so_forward(const so_layer* layers, int32_t L, const double* w, const double* X, int64_t B, double* yhat_out, int threads) {
  if (threads <= 0) threads = omp_get_num_procs();
  int64_t maxw = 1;
  for (int l = 0; l < L - 1; ++l)
    if (layers[l].out > maxw) maxw = layers[l].out;
  double* buf[2] = {NULL, NULL};
  if (L > 1) {
    buf[0] = ws_get(0, (size_t)maxw * B);
    buf[1] = L > 2 ? ws_get(1, (size_t)maxw * B) : NULL;
    if (!buf[0] || (L > 2 && !buf[1])) return -1;
  }
  const double* h = X;
  for (int l = 0; l < L; ++l) {
    double* o = l == L - 1 ? yhat_out : buf[l & 1];
    dense_forward(w + layers[l].w_off, w + layers[l].b_off, h, o, layers[l].out, layers[l].in, B, layers[l].act, threads);
    h = o;
  }
  return 0;
}

/* src/space_inference.jl:90-95  density(z); returns 0 and the log-density in *lp_out (-1: allocation failure) */
int so_logdensity(const so_layer* layers, int32_t L, int64_t N, int32_t M, const double* w_swa, const double* P,
                  const double* X, const double* Y, int32_t out_dim, int64_t B, double sigma_m, const double* z,
                  double* lp_out, int threads) {
  if (threads <= 0) threads = omp_get_num_procs();
  double* w = ws_get(2, (size_t)N);
  double* yhat = ws_get(3, (size_t)out_dim * B);
  if (!w || !yhat) return -1;
  so_reconstruct(w_swa, P, N, M, z, w, threads);
  int rc = so_forward(layers, L, w, X, B, yhat, threads);
  if (rc == 0) {
    const int64_t d = (int64_t)out_dim * B;
    double sse = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : sse) num_threads(threads)
    for (int64_t i = 0; i < d; ++i) {
      const double r = Y[i] - yhat[i];
      sse += r * r;
    }
    /* Distributions 0.24.18 logpdf(MvNormal(mu, sigma::Real), y): -(d*log(2pi) + d*log(sigma^2))/2 - sse/(2 sigma^2) */
    const double c0 = -((double)d * log(2.0 * M_PI) + (double)d * log(sigma_m * sigma_m)) / 2.0;
    *lp_out = c0 - (sse / (sigma_m * sigma_m)) / 2.0;
  }
  return rc;
}
