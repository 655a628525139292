// K2 Gram matrix G = A'A (tall-skinny, fp64 MFMA) and K3 projection P = A*V_M.
// Together with the host eigensolver they replace reference src/subspace_construction.jl:63,65
//     U,s,V = psvd(A);  P = U[:,1:M]*Diagonal(s[1:M])
// through the identity  A = U S V'  =>  A'A = V S^2 V'  and  U[:,1:M]*Diag(s[1:M]) = A*V[:,1:M].
//
// A is N x K column-major with padded leading dimension ldA (multiple of 32 => columns 256-B aligned).
#include <algorithm>
#include <cstdlib>

#include "si_internal.h"

namespace si {

typedef double d4 __attribute__((ext_vector_type(4)));

// two consecutive rows of one column of A as doubles, whatever A is stored in (fp64: the reference; fp32: the opt-in storage of
// SURVEY section 0 Q6 -- half the bytes of the kernels whose time follows the bytes of A; the arithmetic stays fp64)
template <typename AT>
__device__ __forceinline__ double2 load_a2(const AT* p) {
  if constexpr (sizeof(AT) == 8) {
    return *reinterpret_cast<const double2*>(p);
  } else {
    const float2 f = *reinterpret_cast<const float2*>(p);
    return make_double2((double)f.x, (double)f.y);
  }
}

constexpr int GR = 32;   // slab rows
constexpr int GRP = 34;  // padded column stride in LDS (68 dwords): lane l reads column l&15, row 4s + (l>>4), i.e. dword
                         // banks 4*col + 2*row (mod 64) -- distinct inside each 32-lane half for ds_read_b64.  Measured
                         // (SQ_LDS_BANK_CONFLICT): strides 34 / 38 / 42 / 46 all leave the same 19 M conflict cycles of
                         // 59 M active, 36 and 40 are 7-15x worse.

// ------------------------------------------------------------------------------------------------
// K2, K <= 128 (the usual case: K = snapshots collected, 100 at cfg2).  A 32-row slab of ALL columns fits in LDS twice
// (2 x NT*16 columns x 34 doubles: 61 KB at K = 100), so A is read exactly once and the NT(NT+1)/2 upper-triangular
// 16x16 tiles are dealt round-robin to the 8 waves of a 512-thread workgroup: every wave runs a STATIC list of at most
// 5 MFMAs per k step (compile-time tile pairs => accumulators stay in fixed registers).  Two workgroups per CU (4 waves
// per SIMD).  Pipeline per slab, ONE barrier: [k steps 0-3] [ds_write slab s+1 -> other buffer; global loads of slab
// s+2 -> registers] [k steps 4-7] barrier -- the LDS stores and the HBM loads issue under the wave's own MFMAs.  (A
// single-buffered 64-row slab needed two barriers with an exposed store phase between them: the matrix pipe sat at
// 68 %.)  Partial tiles go to Gpart[block][pair][256].
// ------------------------------------------------------------------------------------------------
constexpr int GS_WAVES = 8;

__host__ __device__ constexpr int tri_a(int p, int nt) {
  int a = 0;
  while (p >= nt - a) {
    p -= nt - a;
    ++a;
  }
  return a;
}
__host__ __device__ constexpr int tri_b(int p, int nt) {
  int a = 0;
  while (p >= nt - a) {
    p -= nt - a;
    ++a;
  }
  return a + p;
}

template <int NT, int W>
struct GramWave {
  static constexpr int P = NT * (NT + 1) / 2;
  static constexpr int SLOTS = (P > W) ? (P - W + GS_WAVES - 1) / GS_WAVES : 0;
  static constexpr unsigned mask() {
    unsigned m = 0;
    for (int s = 0; s < SLOTS; ++s) {
      m |= 1u << tri_a(W + GS_WAVES * s, NT);
      m |= 1u << tri_b(W + GS_WAVES * s, NT);
    }
    return m;
  }
  // tile pair of slot SL, forced to compile time (a runtime tri_a() would index the operand registers dynamically)
  template <int SL>
  static __device__ __forceinline__ void mfma_slots(const double (&f)[NT], d4 (&acc)[5]) {
    if constexpr (SL < SLOTS) {
      constexpr int a = tri_a(W + GS_WAVES * SL, NT), b = tri_b(W + GS_WAVES * SL, NT);
      acc[SL] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[a], f[b], acc[SL], 0, 0, 0);
      mfma_slots<SL + 1>(f, acc);
    }
  }
  // fb[s] = element index of (column c, row 4s + q) in buffer 0; OFF = element offset of the buffer in use (compile
  // time, so every read is base register + immediate).  The eight indices are OPAQUE to the compiler (see gram_small_body): reads
  // of neighbouring k steps are 32 B apart and would otherwise be fused into ds_read2_b64, which is banked mod 32 in
  // 16-lane groups -- 2-way conflicts on this column-strided image and a quarter of ds_read_b64's bandwidth
  // (SQ_LDS_BANK_CONFLICT was 46 % of SQ_LDS_IDX_ACTIVE).
  template <int T, int OFF>
  static __device__ __forceinline__ void load_frags(const double* sA, int b, double (&f)[NT]) {
    if constexpr (T < NT) {
      constexpr unsigned M = mask();
      if constexpr ((M >> T) & 1u) f[T] = sA[b + (OFF + T * 16 * GRP)];
      load_frags<T + 1, OFF>(sA, b, f);
    }
  }
  // k steps [S0, S1) of one slab; f[t] = operand of column tile t (same register image serves as A and as B operand)
  template <int S0, int S1, int OFF>
  static __device__ __forceinline__ void steps(const double* sA, const int (&fb)[GR / 4], d4 (&acc)[5]) {
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      double f[NT];
      load_frags<0, OFF>(sA, fb[s], f);
      mfma_slots<0>(f, acc);
    }
  }
  template <int SL>
  static __device__ __forceinline__ void store_slots(double* out, const d4 (&acc)[5], int q, int c) {
    if constexpr (SL < SLOTS) {
      constexpr int p = W + GS_WAVES * SL;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[p * 256 + (q + 4 * r) + 16 * c] = acc[SL][r];  // tile-local (i, j) at i + 16 j
      store_slots<SL + 1>(out, acc, q, c);
    }
  }
  static __device__ __forceinline__ void store(double* out, const d4 (&acc)[5], int q, int c) {
    store_slots<0>(out, acc, q, c);
  }
};

// whole workgroup loop for wave W: the accumulators of a wave never meet another wave's code path, so there is no
// register shuffling at the joins.  All eight paths execute the same barriers (s_barrier counts arrivals per workgroup).
template <int NT, int W, typename AT>
__device__ __forceinline__ void gram_small_body(const AT* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                int col0, double* __restrict__ Gpart, double* sA) {
  constexpr int NC = NT * 16;
  constexpr int NLD = (NC + 31) / 32;  // staging passes: 512 threads cover 32 columns x 32 rows with 16 B per lane
  constexpr int BUF = NC * GRP;        // doubles per LDS buffer
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const int srow = (lane & 15) * 2;            // 16 lanes x 2 rows = one 32-row column
  const int scol0 = W * 4 + (lane >> 4);       // 4 columns per wave per pass
  const AT* Arow = A + srow;
  double2 rg[NLD];
  auto load_slab = [&](int64_t slab) {
    const AT* base = Arow + slab * GR;
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int g = col0 + scol0 + p * 32;
      rg[p] = load_a2<AT>(base + (int64_t)(g < K ? g : K - 1) * ldA);
    }
  };
  auto store_slab = [&](double* dst) {
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int lc = scol0 + p * 32;
      if (lc < NC) *reinterpret_cast<double2*>(dst + lc * GRP + srow) = col0 + lc < K ? rg[p] : make_double2(0.0, 0.0);
    }
  };
  d4 acc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  int fb[GR / 4];
#pragma unroll
  for (int s = 0; s < GR / 4; ++s) {
    fb[s] = c * GRP + q + 4 * s;
    asm volatile("" : "+v"(fb[s]));  // keep the eight indices unrelated for the load/store optimiser (no ds_read2_b64)
  }
  const int64_t nslab = (N + GR - 1) / GR;
  const int64_t stride = gridDim.x;
  int64_t slab = blockIdx.x;
  if (slab >= nslab) {  // (the launcher never starts more blocks than slabs; keep the partial defined anyway)
    GramWave<NT, W>::store(Gpart + (int64_t)blockIdx.x * (NT * (NT + 1) / 2) * 256, acc, q, c);
    return;
  }
  load_slab(slab);
  store_slab(sA);
  if (slab + stride < nslab) load_slab(slab + stride);
  __syncthreads();
  // one slab from buffer BUFI; the other buffer receives slab s+1 in the middle of it
  auto one_slab = [&](auto BUFI) {
    constexpr int bi = decltype(BUFI)::value;
    __builtin_amdgcn_s_setprio(1);
    GramWave<NT, W>::template steps<0, GR / 8, bi * BUF>(sA, fb, acc);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (slab + stride < nslab) {
      store_slab(sA + (bi ^ 1) * BUF);                          // free since the barrier of the last iteration
      if (slab + 2 * stride < nslab) load_slab(slab + 2 * stride);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    GramWave<NT, W>::template steps<GR / 8, GR / 4, bi * BUF>(sA, fb, acc);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    slab += stride;
  };
  while (slab < nslab) {
    one_slab(std::integral_constant<int, 0>{});
    if (slab < nslab) one_slab(std::integral_constant<int, 1>{});
  }
  GramWave<NT, W>::store(Gpart + (int64_t)blockIdx.x * (NT * (NT + 1) / 2) * 256, acc, q, c);
}

template <int NT, typename AT>
#ifndef GS_MINW
#define GS_MINW 4
#endif
__global__ __launch_bounds__(512, GS_MINW) void gram_small_kernel(const AT* __restrict__ A, int64_t ldA, int64_t N,
                                                                  int K, int col0, double* __restrict__ Gpart) {
  extern __shared__ double sA[];  // [2][NT*16][GRP]
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: gram_small_body<NT, 0, AT>(A, ldA, N, K, col0, Gpart, sA); break;
    case 1: gram_small_body<NT, 1, AT>(A, ldA, N, K, col0, Gpart, sA); break;
    case 2: gram_small_body<NT, 2, AT>(A, ldA, N, K, col0, Gpart, sA); break;
    case 3: gram_small_body<NT, 3, AT>(A, ldA, N, K, col0, Gpart, sA); break;
    case 4: gram_small_body<NT, 4, AT>(A, ldA, N, K, col0, Gpart, sA); break;
    case 5: gram_small_body<NT, 5, AT>(A, ldA, N, K, col0, Gpart, sA); break;
    case 6: gram_small_body<NT, 6, AT>(A, ldA, N, K, col0, Gpart, sA); break;
    default: gram_small_body<NT, 7, AT>(A, ldA, N, K, col0, Gpart, sA); break;
  }
}

// sums the per-block partial tiles in a fixed order (bit-reproducible), writes both triangles of G.  1024 threads:
// element e of the tile x 4 lanes; lane j adds partials j, j+4, ... in order, the four lane sums are added in a fixed tree.
__global__ __launch_bounds__(1024) void gram_small_reduce_kernel(const double* __restrict__ Gpart, int nblocks, int nt,
                                                                 int K, int col0, double* __restrict__ G) {
  __shared__ double red[4][256];
  const int p = blockIdx.x;  // tile pair
  const int a = tri_a(p, nt), b = tri_b(p, nt);
  const int e = threadIdx.x & 255, ln = threadIdx.x >> 8, i = e & 15, j = e >> 4;
  const int gi = col0 + a * 16 + i, gj = col0 + b * 16 + j;
  const int64_t stride = (int64_t)(nt * (nt + 1) / 2) * 256;
  const double* src = Gpart + (int64_t)p * 256 + e;
  double s = 0.0;
  int sp = ln;
  for (; sp + 28 < nblocks; sp += 32) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sp + 4 * u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sp < nblocks; sp += 4) s += src[(int64_t)sp * stride];
  red[ln][e] = s;
  __syncthreads();
  if (ln == 0 && gi < K && gj < K) {
    s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    if (a != b) {
      G[gi + (int64_t)K * gj] = s;
      G[gj + (int64_t)K * gi] = s;
    } else if (i <= j) {  // diagonal tile: take the upper triangle, mirror it (exactly symmetric output)
      G[gi + (int64_t)K * gj] = s;
      G[gj + (int64_t)K * gi] = s;
    }
  }
}

template <int NT, typename AT>
static void launch_gram_small(hipStream_t st, const AT* A, int64_t ldA, int64_t N, int K, int col0, double* Gpart,
                              double* G, int nblocks, Ctx* prof, double flops, double bytes) {
  constexpr size_t lds = (size_t)2 * NT * 16 * GRP * sizeof(double);
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(gram_small_kernel<NT, AT>), lds);
  {
    ProfScope ps(prof, SI_K_GRAM, flops, bytes);
    hipLaunchKernelGGL((gram_small_kernel<NT, AT>), dim3(nblocks), dim3(512), lds, st, A, ldA, N, K, col0, Gpart);
  }
  {
    ProfScope ps(prof, SI_K_GRAM_RED, 0.0, (double)nblocks * NT * (NT + 1) / 2 * 256 * 8.0);
    hipLaunchKernelGGL(gram_small_reduce_kernel, dim3(NT * (NT + 1) / 2), dim3(1024), 0, st, Gpart, nblocks, NT, K, col0, G);
  }
}

// ------------------------------------------------------------------------------------------------
// K2, K > 128: the columns are cut into 128-wide panels.  Diagonal panel pairs use gram_small_kernel on the panel's
// columns; an off-diagonal pair (I < J) stages a 32-row slab of BOTH panels (up to 256 columns x 34 doubles = 70 KB:
// two workgroups per CU) and wave w multiplies tile row w of panel I with all NTJ tiles of panel J:
//     D[i][j] += sum_r A[r][ci0 + 16w + i] * A[r][cj0 + 16b + j],  b = 0..NTJ-1     (NTJ MFMAs per 1 + NTJ operand reads)
// Every panel pair is read once per pair it takes part in, the tile count is the minimal one.
// ------------------------------------------------------------------------------------------------
constexpr int GR2 = 32;   // slab rows of the off-diagonal kernel
constexpr int GRP2 = 34;  // 68 dwords per column: operand reads touch banks 4c + 2q (mod 64), conflict-free

template <int NTJ, typename AT>
__global__ __launch_bounds__(512, 4) void gram_off_kernel(const AT* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                          int ci0, int cj0, double* __restrict__ Gpart) {
  extern __shared__ double sA[];  // [(8 + NTJ) * 16][GRP2]: panel I columns first, then panel J
  constexpr int NC = (8 + NTJ) * 16;
  constexpr int NLD = (NC + 31) / 32;  // 512 threads x 16 B cover 32 columns x 32 rows per pass
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c = lane & 15;
  const int srow = (tid & 15) * 2;  // 16 lanes x 2 rows = one 32-row column
  const int scol0 = tid >> 4;       // 0..31
  const AT* Arow = A + srow;
  double2 rg[NLD];
  auto gcol = [&](int lc) { return lc < 128 ? ci0 + lc : cj0 + (lc - 128); };
  auto load_slab = [&](int64_t slab) {
    const AT* base = Arow + slab * GR2;
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int lc = scol0 + p * 32;
      const int g = gcol(lc < NC ? lc : 0);
      rg[p] = load_a2<AT>(base + (int64_t)(g < K ? g : K - 1) * ldA);
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int lc = scol0 + p * 32;
      if (lc < NC) *reinterpret_cast<double2*>(sA + lc * GRP2 + srow) = gcol(lc) < K ? rg[p] : make_double2(0.0, 0.0);
    }
  };
  d4 acc[NTJ];
#pragma unroll
  for (int b = 0; b < NTJ; ++b) acc[b] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* pa = sA + (wave * 16 + c) * GRP2 + q;
  const double* pb = sA + (128 + c) * GRP2 + q;
  const int64_t nslab = (N + GR2 - 1) / GR2;
  int64_t slab = blockIdx.x;
  if (slab < nslab) load_slab(slab);
  for (; slab < nslab; slab += gridDim.x) {
    store_slab();
    __syncthreads();
    const int64_t next = slab + gridDim.x;
    if (next < nslab) load_slab(next);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int s = 0; s < GR2 / 4; ++s) {
      const double fa = pa[4 * s];
      double fb[NTJ];
#pragma unroll
      for (int b = 0; b < NTJ; ++b) fb[b] = pb[b * 16 * GRP2 + 4 * s];
#pragma unroll
      for (int b = 0; b < NTJ; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[b], acc[b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  // partial tile (a = wave, b): element (i, j) at i + 16 j, tiles ordered a * NTJ + b
  double* out = Gpart + ((int64_t)blockIdx.x * 8 * NTJ + wave * NTJ) * 256;
#pragma unroll
  for (int b = 0; b < NTJ; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[b * 256 + (q + 4 * r) + 16 * c] = acc[b][r];
}

__global__ __launch_bounds__(256) void gram_off_reduce_kernel(const double* __restrict__ Gpart, int nblocks, int ntj, int K,
                                                              int ci0, int cj0, double* __restrict__ G) {
  const int t = blockIdx.x;  // tile a * ntj + b
  const int a = t / ntj, b = t % ntj;
  const int e = threadIdx.x, i = e & 15, j = e >> 4;
  const int gi = ci0 + a * 16 + i, gj = cj0 + b * 16 + j;
  const int64_t stride = (int64_t)8 * ntj * 256;
  const double* src = Gpart + (int64_t)t * 256 + e;
  double s = 0.0;
  int sp = 0;
  for (; sp + 8 <= nblocks; sp += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sp + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sp < nblocks; ++sp) s += src[(int64_t)sp * stride];
  if (gi < K && gj < K) {
    G[gi + (int64_t)K * gj] = s;
    G[gj + (int64_t)K * gi] = s;
  }
}

template <int NTJ, typename AT>
static void launch_gram_off(hipStream_t st, const AT* A, int64_t ldA, int64_t N, int K, int ci0, int cj0, double* Gpart,
                            double* G, int nblocks, Ctx* prof, double flops, double bytes) {
  constexpr size_t lds = (size_t)(8 + NTJ) * 16 * GRP2 * sizeof(double);
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(gram_off_kernel<NTJ, AT>), lds);
  {
    ProfScope ps(prof, SI_K_GRAM, flops, bytes);
    hipLaunchKernelGGL((gram_off_kernel<NTJ, AT>), dim3(nblocks), dim3(512), lds, st, A, ldA, N, K, ci0, cj0, Gpart);
  }
  {
    ProfScope ps(prof, SI_K_GRAM_RED, 0.0, (double)nblocks * 8 * NTJ * 256 * 8.0);
    hipLaunchKernelGGL(gram_off_reduce_kernel, dim3(8 * NTJ), dim3(256), 0, st, Gpart, nblocks, NTJ, K, ci0, cj0, G);
  }
}

template <typename AT>
static void gram_diag_dispatch(hipStream_t st, const AT* A, int64_t ldA, int64_t N, int K, int col0, int nt, double* Gpart,
                               double* G, int nblocks, Ctx* prof, double flops, double bytes) {
  switch (nt) {
    case 1: launch_gram_small<1, AT>(st, A, ldA, N, K, col0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 2: launch_gram_small<2, AT>(st, A, ldA, N, K, col0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 3: launch_gram_small<3, AT>(st, A, ldA, N, K, col0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 4: launch_gram_small<4, AT>(st, A, ldA, N, K, col0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 5: launch_gram_small<5, AT>(st, A, ldA, N, K, col0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 6: launch_gram_small<6, AT>(st, A, ldA, N, K, col0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 7: launch_gram_small<7, AT>(st, A, ldA, N, K, col0, Gpart, G, nblocks, prof, flops, bytes); break;
    default: launch_gram_small<8, AT>(st, A, ldA, N, K, col0, Gpart, G, nblocks, prof, flops, bytes); break;
  }
}

template <typename AT>
static void gram_off_dispatch(hipStream_t st, const AT* A, int64_t ldA, int64_t N, int K, int ci0, int cj0, int ntj,
                              double* Gpart, double* G, int nblocks, Ctx* prof, double flops, double bytes) {
  switch (ntj) {
    case 1: launch_gram_off<1, AT>(st, A, ldA, N, K, ci0, cj0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 2: launch_gram_off<2, AT>(st, A, ldA, N, K, ci0, cj0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 3: launch_gram_off<3, AT>(st, A, ldA, N, K, ci0, cj0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 4: launch_gram_off<4, AT>(st, A, ldA, N, K, ci0, cj0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 5: launch_gram_off<5, AT>(st, A, ldA, N, K, ci0, cj0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 6: launch_gram_off<6, AT>(st, A, ldA, N, K, ci0, cj0, Gpart, G, nblocks, prof, flops, bytes); break;
    case 7: launch_gram_off<7, AT>(st, A, ldA, N, K, ci0, cj0, Gpart, G, nblocks, prof, flops, bytes); break;
    default: launch_gram_off<8, AT>(st, A, ldA, N, K, ci0, cj0, Gpart, G, nblocks, prof, flops, bytes); break;
  }
}

// ------------------------------------------------------------------------------------------------
// K2, K <= 208: ONE pass over A by the LDS-DMA kernel of kernels_gram_wave.hip (one 4-wave workgroup per CU), then a
// two-stage fixed-order sum of the per-workgroup partial tiles (bit-reproducible): stage 1 gives tile pair p to GR2_Y
// workgroups, each adding 1/GR2_Y of the partials; stage 2 adds the GR2_Y sums in order and writes both triangles of G.
// (The one-stage reduction of round 1 gave a tile to ONE workgroup: 28 workgroups pulling 29 MB at K = 100 took 0.05 ms,
// a fifth of the Gram kernel itself.)
// ------------------------------------------------------------------------------------------------
int gram_wave_blocks_per_cu(int nt);
bool launch_gram_wave_part0(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks);
bool launch_gram_wave_part1(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks);
bool launch_gram_wave_part2(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks);

constexpr int GR2_Y = 16;
__global__ __launch_bounds__(256) void gram_reduce_stage1_kernel(const double* __restrict__ Gpart, int nblocks, int P,
                                                                 double* __restrict__ scr) {
  const int p = blockIdx.x, y = blockIdx.y, e = threadIdx.x;
  const int64_t stride = (int64_t)P * 256;
  const int per = (nblocks + GR2_Y - 1) / GR2_Y;
  const int sp0 = y * per, sp1 = sp0 + per < nblocks ? sp0 + per : nblocks;
  const double* src = Gpart + (int64_t)p * 256 + e;
  double s = 0.0;
  int sp = sp0;
  for (; sp + 8 <= sp1; sp += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sp + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sp < sp1; ++sp) s += src[(int64_t)sp * stride];
  scr[((int64_t)y * P + p) * 256 + e] = s;
}
__global__ __launch_bounds__(256) void gram_reduce_stage2_kernel(const double* __restrict__ scr, int nt, int K,
                                                                 double* __restrict__ G) {
  const int p = blockIdx.x, e = threadIdx.x;
  const int P = nt * (nt + 1) / 2;
  double t = 0.0;
#pragma unroll
  for (int yy = 0; yy < GR2_Y; ++yy) t += scr[((int64_t)yy * P + p) * 256 + e];
  const int a = tri_a(p, nt), b = tri_b(p, nt);
  const int i = e & 15, j = e >> 4;
  const int gi = a * 16 + i, gj = b * 16 + j;
  if (gi < K && gj < K && (a != b || i <= j)) {  // diagonal tile: take the upper triangle, mirror it (exactly symmetric output)
    G[gi + (int64_t)K * gj] = t;
    G[gj + (int64_t)K * gi] = t;
  }
}

static bool gram_wave_dispatch(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* Gpart, double* G, int nblocks,
                               Ctx* prof, double flops, double bytes) {
  const int nt = (K + 15) / 16, P = nt * (nt + 1) / 2;
  double* scr = Gpart + (size_t)nblocks * P * 256;
  {
    ProfScope ps(prof, SI_K_GRAM, flops, bytes);
    if (!(launch_gram_wave_part0(st, A, ldA, N, K, Gpart, nblocks) || launch_gram_wave_part1(st, A, ldA, N, K, Gpart, nblocks) ||
          launch_gram_wave_part2(st, A, ldA, N, K, Gpart, nblocks)))
      return false;
  }
  ProfScope ps(prof, SI_K_GRAM_RED, 0.0, (double)nblocks * P * 256 * 8.0);
  hipLaunchKernelGGL(gram_reduce_stage1_kernel, dim3(P, GR2_Y), dim3(256), 0, st, Gpart, nblocks, P, scr);
  hipLaunchKernelGGL(gram_reduce_stage2_kernel, dim3(P), dim3(256), 0, st, scr, nt, K, G);
  return true;
}

#ifdef SI_DEV_KNOBS   // development build (tools/gram_bench.hip): SI_GRAM_PANELS=1 forces the 128-column panel kernels for every K
static bool gram_force_panels() {
  static const bool v = [] {
    const char* e = getenv("SI_GRAM_PANELS");
    return e && e[0] == '1';
  }();
  return v;
}
#else
static constexpr bool gram_force_panels() { return false; }
#endif

template <typename AT>
static void gram_panels(hipStream_t st, const AT* A, int64_t ldA, int64_t N, int64_t K, double* Gpart, double* G, int64_t nb_diag,
                        int64_t nb_off, int npan, Ctx* prof, double tot_flops, double tot_bytes) {
  bool first = true;
  for (int I = 0; I < npan; ++I) {
    const int ci0 = I * 128;
    const int nti = (int)((std::min<int64_t>(K, ci0 + 128) - ci0 + 15) / 16);
    // the algorithmic flops / bytes of the whole Gram are booked on the first launch, the rest add time only
    gram_diag_dispatch<AT>(st, A, ldA, N, (int)K, ci0, nti, Gpart, G, (int)nb_diag, prof, first ? tot_flops : 0.0,
                           first ? tot_bytes : 0.0);
    first = false;
    for (int J = I + 1; J < npan; ++J) {
      const int cj0 = J * 128;
      const int ntj = (int)((std::min<int64_t>(K, cj0 + 128) - cj0 + 15) / 16);
      gram_off_dispatch<AT>(st, A, ldA, N, (int)K, ci0, cj0, ntj, Gpart, G, (int)nb_off, prof, 0.0, 0.0);
    }
  }
}

// a_dtype = SI_F32: A holds floats (ldA in elements); the register-staged panel kernels widen them on the way to LDS
size_t launch_gram(hipStream_t st, const void* Av, int64_t ldA, int64_t N, int64_t K, double* Gpart,
                   double* G, int num_cu, Ctx* prof, int32_t a_dtype) {
  const double* A = static_cast<const double*>(Av);
  // workspace: the largest single launch (an off-diagonal pair of full panels: 64 tiles per block)
  const int64_t nslab_d = (N + GR - 1) / GR, nslab_o = (N + GR2 - 1) / GR2;
  int64_t nb_diag = std::min<int64_t>((int64_t)num_cu * 2, nslab_d);
  int64_t nb_off = std::min<int64_t>((int64_t)num_cu * 2, nslab_o);
  const int npan = (int)((K + 127) / 128);
  const int ntw = (int)((K + 15) / 16);
  const bool wave_path = ntw <= 13 && !gram_force_panels() && a_dtype != SI_F32;
  const int64_t nb_wave = std::min<int64_t>((int64_t)num_cu * gram_wave_blocks_per_cu(ntw), nslab_d);
  const size_t need = wave_path ? ((size_t)nb_wave + GR2_Y) * (size_t)(ntw * (ntw + 1) / 2) * 256 * sizeof(double)
                                : (size_t)std::max<int64_t>(nb_diag * 36, npan > 1 ? nb_off * 64 : 0) * 256 * sizeof(double);
  if (Gpart == nullptr) return need;
  const double tot_flops = (double)N * (double)K * (double)(K + 1), tot_bytes = (double)N * (double)K * (a_dtype == SI_F32 ? 4.0 : 8.0);
  if (wave_path && gram_wave_dispatch(st, A, ldA, N, (int)K, Gpart, G, (int)nb_wave, prof, tot_flops, tot_bytes)) return need;
  if (a_dtype == SI_F32)
    gram_panels<float>(st, static_cast<const float*>(Av), ldA, N, K, Gpart, G, nb_diag, nb_off, npan, prof, tot_flops, tot_bytes);
  else
    gram_panels<double>(st, A, ldA, N, K, Gpart, G, nb_diag, nb_off, npan, prof, tot_flops, tot_bytes);
  return need;
}

// ------------------------------------------------------------------------------------------------
// K3.  P = A * V_M  (N x K times K x M).  HBM-bound (intensity K*M/(4(K+M)) flop/B), so plain fp64 VALU
// FMAs: each thread owns two consecutive rows (16-B loads, lanes = consecutive rows => 1 KiB contiguous per
// wave per column of A) and MT accumulators per row; V[k][m] is wave-uniform and comes through scalar loads.
// Algorithmic bytes: N*(K+M)*8.
// ------------------------------------------------------------------------------------------------
template <int MT, typename AT>
__global__ __launch_bounds__(256) void project_kernel(const AT* __restrict__ A, int64_t ldA, int64_t N,
                                                      int K, const double* __restrict__ V, int Mpad,
                                                      int m0, int M, double* __restrict__ P, int64_t ldP) {
  const int64_t npair = (N + 1) >> 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += stride) {
    const int64_t r = p << 1;
    double2 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = make_double2(0.0, 0.0);
#pragma unroll 2
    for (int k = 0; k < K; ++k) {
      const double2 av = load_a2<AT>(A + r + (int64_t)k * ldA);
      const double* vk = V + (int64_t)k * Mpad + m0;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const double v = vk[j];
        acc[j].x = fma(av.x, v, acc[j].x);
        acc[j].y = fma(av.y, v, acc[j].y);
      }
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      if (m0 + j < M) *reinterpret_cast<double2*>(P + r + (int64_t)(m0 + j) * ldP) = acc[j];
    }
  }
}

#ifdef SI_DEV_KNOBS   // development build: SI_PROJECT_GEMM=1 keeps the generic GEMM for wide subspaces (comparison runs)
static bool project_force_gemm() {
  static const bool v = [] {
    const char* e = getenv("SI_PROJECT_GEMM");
    return e && e[0] == '1';
  }();
  return v;
}
#else
static constexpr bool project_force_gemm() { return false; }
#endif

int project_mpad(int M) {
  int m0 = 0;
  while (m0 < M) {
    const int rem = M - m0;
    m0 += rem > 24 ? 32 : rem > 16 ? 24 : rem > 8 ? 16 : 8;
  }
  return m0;
}

template <typename AT>
static void project_valu(hipStream_t st, const AT* A, int64_t ldA, int64_t N, int64_t K, const double* V, int32_t M, int32_t Mpad,
                         double* P, int64_t ldP, int num_cu) {
  int64_t blocks = (((N + 1) >> 1) + 255) / 256;
  if (blocks > (int64_t)num_cu * 8) blocks = (int64_t)num_cu * 8;
  if (blocks < 1) blocks = 1;
  int m0 = 0;
  while (m0 < M) {
    const int rem = M - m0;
    if (rem > 24) {
      hipLaunchKernelGGL((project_kernel<32, AT>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP);
      m0 += 32;
    } else if (rem > 16) {
      hipLaunchKernelGGL((project_kernel<24, AT>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP);
      m0 += 24;
    } else if (rem > 8) {
      hipLaunchKernelGGL((project_kernel<16, AT>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP);
      m0 += 16;
    } else {
      hipLaunchKernelGGL((project_kernel<8, AT>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP);
      m0 += 8;
    }
  }
}

void launch_project(hipStream_t st, const void* Av, int64_t ldA, int64_t N, int64_t K, const double* V,
                    int32_t M, int32_t Mpad, double* P, int64_t ldP, int num_cu, int32_t a_dtype) {
  if (a_dtype == SI_F32) {   // fp32-stored A: half the bytes per row of A
    const float* A32 = static_cast<const float*>(Av);
    // wide subspace: one pass on the matrix cores (kernels_project.hip, fp32 slabs widened on the operand read); else the
    // streaming VALU kernel in passes of <= 32 columns
    if (M > 32 && launch_project_stream_f32(st, A32, ldA, N, K, V, M, Mpad, P, ldP, num_cu)) return;
    project_valu<float>(st, A32, ldA, N, K, V, M, Mpad, P, ldP, num_cu);
    return;
  }
  const double* A = static_cast<const double*>(Av);
  if (M > 32) {  // wide subspace: one pass over A on the matrix cores instead of ceil(M/32) VALU passes
    if (!project_force_gemm() && launch_project_stream(st, A, ldA, N, K, V, M, Mpad, P, ldP, num_cu)) return;  // K <= 128: slab stream
    if (N < 0x7fffffff) {
      launch_project_mfma(st, A, ldA, N, K, V, M, Mpad, P, ldP);
      return;
    }
  }
  project_valu<double>(st, A, ldA, N, K, V, M, Mpad, P, ldP, num_cu);
}

}  // namespace si
