// K2 Gram matrix G = A'A (tall-skinny, fp64 MFMA) and K3 projection P = A*V_M.
// Together with the host eigensolver they replace reference src/subspace_construction.jl:63,65
//     U,s,V = psvd(A);  P = U[:,1:M]*Diagonal(s[1:M])
// through the identity  A = U S V'  =>  A'A = V S^2 V'  and  U[:,1:M]*Diag(s[1:M]) = A*V[:,1:M].
//
// A is N x K column-major with padded leading dimension ldA (multiple of 32 => columns 256-B aligned).
#include "si_internal.h"

namespace si {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// K2.  The K columns are cut into panels of 64; a workgroup owns one upper-triangular panel pair (I <= J)
// and a strided set of 64-row slabs.  A slab of both panels is staged in LDS as sA[col][row] with a row
// stride of 66 doubles: the MFMA operand read (lane l -> column l&15, row 4s + (l>>4)) then touches
// dword banks 4*col + 2*row (mod 64): 32 distinct bank pairs per 32-lane half, conflict-free.
// Wave w owns tile row a=(w+rot)&3 of the 4x4 tile block (rotated per workgroup so that the lighter rows
// of a diagonal pair do not always sit on the same SIMD) and keeps 4 accumulators:
//     D[i][j] += sum_r A[r][64I+16a+i] * A[r][64J+16b+j]       v_mfma_f64_16x16x4_f64, k = slab rows
// Partials go to a slab per (split, pair) and are summed in fixed order by gram_reduce_kernel, so the
// result is bit-reproducible.  Algorithmic bytes: N*K*8 (A read once); flops N*K*(K+1).
// ------------------------------------------------------------------------------------------------
constexpr int GP = 64;   // panel width (columns)
constexpr int GR = 64;   // slab rows
constexpr int GRP = 66;  // padded row stride in LDS

__device__ __forceinline__ void pair_from_index(int p, int np, int& I, int& J) {
  // enumerate (I,J), I <= J < np, row-major
  int i = 0;
  while (p >= np - i) {
    p -= np - i;
    ++i;
  }
  I = i;
  J = i + p;
}

// Staging is branch-free: the rows of A between N and ldA are zero (si_construct_begin clears A and K1 never writes
// them; ldA is a multiple of the 64-row slab), column indices past K are clamped to a legal column and zeroed when the
// registers go to LDS.  The next slab is loaded into registers while the MFMAs of the current one run.
template <int B0>
__device__ __forceinline__ void gram_steps(const double* pa, const double* pb, d4 (&acc)[4]) {
#pragma unroll 4
  for (int s = 0; s < GR / 4; ++s) {
    const double fa = pa[4 * s];
    double fb[4];
#pragma unroll
    for (int b = B0; b < 4; ++b) fb[b] = pb[b * 16 * GRP + 4 * s];
#pragma unroll
    for (int b = B0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[b], acc[b], 0, 0, 0);
  }
}

__global__ __launch_bounds__(256, 2) void gram_pair_kernel(const double* __restrict__ A, int64_t ldA,
                                                           int64_t N, int K, int npanels,
                                                           double* __restrict__ Gpart) {
  extern __shared__ double sA[];  // 2 * GP * GRP doubles (67.6 KB: above the static limit)
  const int pair = blockIdx.x;
  int I, J;
  pair_from_index(pair, npanels, I, J);
  const bool diag = (I == J);
  double* sI = sA;
  double* sJ = diag ? sA : sA + GP * GRP;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the per-tile branches below are s_cbranch
  const int a = (wave + pair + blockIdx.y) & 3;
  const int q = lane >> 4, c = lane & 15;

  d4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = (d4){0.0, 0.0, 0.0, 0.0};

  const int64_t nslab = (N + GR - 1) / GR;
  // staging map: a wave-instruction covers 2 columns x 64 rows with 16 B per lane
  const int srow = (lane & 31) * 2;          // even row inside the slab
  const int scol0 = wave * 2 + (lane >> 5);  // column inside the panel, step 8 per pass
  constexpr int NP = GP / 8;
  const double* Arow = A + srow;
  const int cI0 = I * GP + scol0, cJ0 = J * GP + scol0;
  double2 rI[NP], rJ[NP];
  // column indices past K are clamped to K-1 (legal address) and the value is zeroed when it goes to LDS; the
  // addresses are recomputed per load (a few VALU ops hidden under 64-cycle MFMAs) instead of held in 32 registers
  auto load_slab = [&](int64_t slab) {
    const double* base = Arow + slab * GR;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int g = cI0 + p * 8;
      rI[p] = *reinterpret_cast<const double2*>(base + (int64_t)(g < K ? g : K - 1) * ldA);
    }
    if (!diag) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int g = cJ0 + p * 8;
        rJ[p] = *reinterpret_cast<const double2*>(base + (int64_t)(g < K ? g : K - 1) * ldA);
      }
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = scol0 + p * 8;
      *reinterpret_cast<double2*>(sI + col * GRP + srow) = (cI0 + p * 8 < K) ? rI[p] : make_double2(0.0, 0.0);
    }
    if (!diag) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int col = scol0 + p * 8;
        *reinterpret_cast<double2*>(sJ + col * GRP + srow) = (cJ0 + p * 8 < K) ? rJ[p] : make_double2(0.0, 0.0);
      }
    }
  };

  const double* pa = sI + (a * 16 + c) * GRP + q;
  const double* pb = sJ + c * GRP + q;
  const int b0 = diag ? a : 0;
  int64_t slab = blockIdx.y;
  if (slab < nslab) load_slab(slab);
  for (; slab < nslab; slab += gridDim.y) {
    store_slab();
    __syncthreads();
    const int64_t next = slab + gridDim.y;
    if (next < nslab) load_slab(next);  // in flight under the MFMAs below
    // tiles b0..3 of this wave's tile row (b0 = a on a diagonal pair, else 0).  The four cases are separate code
    // paths with a STATIC set of MFMAs: a per-MFMA condition would make the compiler copy the 8-register
    // accumulators around every instruction (measured: 300 v_mov_b64 per k loop, 10x slower).
    __builtin_amdgcn_s_setprio(1);
    switch (b0) {
      case 0: gram_steps<0>(pa, pb, acc); break;
      case 1: gram_steps<1>(pa, pb, acc); break;
      case 2: gram_steps<2>(pa, pb, acc); break;
      default: gram_steps<3>(pa, pb, acc); break;
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();  // every wave is done reading before the next slab overwrites the buffer
  }

  // partial block, column-major 64x64: element (ii, jj) at ii + 64*jj
  const int npairs = gridDim.x;
  double* out = Gpart + ((int64_t)blockIdx.y * npairs + pair) * (GP * GP);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (diag && b < a) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ii = a * 16 + q + 4 * r;
      const int jj = b * 16 + c;
      out[ii + GP * jj] = acc[b][r];
    }
  }
}

// sums the per-split partial blocks in split order (bit-reproducible) and mirrors the upper triangle
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ Gpart, int nsplit,
                                                          int npanels, int K, double* __restrict__ G) {
  const int pair = blockIdx.x;
  const int npairs = gridDim.x;
  int I, J;
  pair_from_index(pair, npanels, I, J);
  const int e = blockIdx.y * 256 + threadIdx.x;  // element of the 64x64 block
  const int ii = e % GP, jj = e / GP;
  if (I == J && (ii >> 4) > (jj >> 4)) return;  // tile below the diagonal: never computed
  const int gi = I * GP + ii, gj = J * GP + jj;
  if (gi >= K || gj >= K) return;
  const double* src = Gpart + (int64_t)pair * (GP * GP) + e;
  const int64_t stride = (int64_t)npairs * (GP * GP);
  double s = 0.0;
  int sp = 0;
  for (; sp + 8 <= nsplit; sp += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(sp + u) * stride];  // 8 independent loads in flight
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; sp < nsplit; ++sp) s += src[(int64_t)sp * stride];
  G[gi + (int64_t)K * gj] = s;
  if (!(I == J && (ii >> 4) == (jj >> 4))) G[gj + (int64_t)K * gi] = s;  // diagonal tiles hold both triangles
}

// ------------------------------------------------------------------------------------------------
// K2, K <= 128 (the usual case: K = snapshots collected, 100 at cfg2).  A 64-row slab of ALL columns fits in LDS
// (NT*16 columns x 66 doubles), so A is read exactly once and the NT(NT+1)/2 upper-triangular 16x16 tiles are dealt
// round-robin to the 8 waves of a 512-thread workgroup: every wave runs a STATIC list of at most 5 MFMAs per k step
// (compile-time tile pairs => accumulators stay in fixed registers).  Two workgroups per CU (4 waves per SIMD); the
// next slab is prefetched into registers under the MFMAs.  Partial tiles go to Gpart[block][pair][256].
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
  template <int T>
  static __device__ __forceinline__ void load_frags(const double* base, int s, double (&f)[NT]) {
    if constexpr (T < NT) {
      constexpr unsigned M = mask();
      if constexpr ((M >> T) & 1u) f[T] = base[T * 16 * GRP + 4 * s];
      load_frags<T + 1>(base, s, f);
    }
  }
  // one 64-row slab: 16 k steps; f[t] = operand of column tile t (same register image serves as A and as B operand)
  static __device__ __forceinline__ void steps(const double* base, d4 (&acc)[5]) {
#pragma unroll 2
    for (int s = 0; s < GR / 4; ++s) {
      double f[NT];
      load_frags<0>(base, s, f);
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
template <int NT, int W>
__device__ __forceinline__ void gram_small_body(const double* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                double* __restrict__ Gpart, double* sA) {
  constexpr int NC = NT * 16;
  constexpr int NLD = (NC + 15) / 16;  // staging passes: 512 threads cover 16 columns x 64 rows with 16 B per lane
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = lane >> 4, c = lane & 15;
  const int srow = (lane & 31) * 2;
  const int scol0 = W * 2 + (lane >> 5);
  const double* Arow = A + srow;
  double2 rg[NLD];
  auto load_slab = [&](int64_t slab) {
    const double* base = Arow + slab * GR;
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int g = scol0 + p * 16;
      rg[p] = *reinterpret_cast<const double2*>(base + (int64_t)(g < K ? g : K - 1) * ldA);
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int g = scol0 + p * 16;
      if (g < NC) *reinterpret_cast<double2*>(sA + g * GRP + srow) = g < K ? rg[p] : make_double2(0.0, 0.0);
    }
  };
  d4 acc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* fbase = sA + c * GRP + q;
  const int64_t nslab = (N + GR - 1) / GR;
  int64_t slab = blockIdx.x;
  if (slab < nslab) load_slab(slab);
  for (; slab < nslab; slab += gridDim.x) {
    store_slab();
    __syncthreads();
    const int64_t next = slab + gridDim.x;
    if (next < nslab) load_slab(next);
    __builtin_amdgcn_s_setprio(1);
    GramWave<NT, W>::steps(fbase, acc);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  GramWave<NT, W>::store(Gpart + (int64_t)blockIdx.x * (NT * (NT + 1) / 2) * 256, acc, q, c);
}

template <int NT>
#ifndef GS_MINW
#define GS_MINW 4
#endif
__global__ __launch_bounds__(512, GS_MINW) void gram_small_kernel(const double* __restrict__ A, int64_t ldA, int64_t N,
                                                                  int K, double* __restrict__ Gpart) {
  extern __shared__ double sA[];  // [NT*16][GRP]
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: gram_small_body<NT, 0>(A, ldA, N, K, Gpart, sA); break;
    case 1: gram_small_body<NT, 1>(A, ldA, N, K, Gpart, sA); break;
    case 2: gram_small_body<NT, 2>(A, ldA, N, K, Gpart, sA); break;
    case 3: gram_small_body<NT, 3>(A, ldA, N, K, Gpart, sA); break;
    case 4: gram_small_body<NT, 4>(A, ldA, N, K, Gpart, sA); break;
    case 5: gram_small_body<NT, 5>(A, ldA, N, K, Gpart, sA); break;
    case 6: gram_small_body<NT, 6>(A, ldA, N, K, Gpart, sA); break;
    default: gram_small_body<NT, 7>(A, ldA, N, K, Gpart, sA); break;
  }
}

// sums the per-block partial tiles in block order (bit-reproducible), writes both triangles of G
__global__ __launch_bounds__(256) void gram_small_reduce_kernel(const double* __restrict__ Gpart, int nblocks, int nt,
                                                                int K, double* __restrict__ G) {
  const int p = blockIdx.x;  // tile pair
  const int a = tri_a(p, nt), b = tri_b(p, nt);
  const int e = threadIdx.x, i = e & 15, j = e >> 4;
  const int gi = a * 16 + i, gj = b * 16 + j;
  const int64_t stride = (int64_t)(nt * (nt + 1) / 2) * 256;
  const double* src = Gpart + (int64_t)p * 256 + e;
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
    if (a != b) {
      G[gi + (int64_t)K * gj] = s;
      G[gj + (int64_t)K * gi] = s;
    } else if (i <= j) {  // diagonal tile: take the upper triangle, mirror it (exactly symmetric output)
      G[gi + (int64_t)K * gj] = s;
      G[gj + (int64_t)K * gi] = s;
    }
  }
}

template <int NT>
static void launch_gram_small(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* Gpart, double* G,
                              int nblocks, Ctx* prof) {
  constexpr size_t lds = (size_t)NT * 16 * GRP * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gram_small_kernel<NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  {
    ProfScope ps(prof, SI_K_GRAM, (double)N * (double)K * (double)(K + 1), (double)N * (double)K * 8.0);
    hipLaunchKernelGGL(gram_small_kernel<NT>, dim3(nblocks), dim3(512), lds, st, A, ldA, N, K, Gpart);
  }
  {
    ProfScope ps(prof, SI_K_GRAM_RED, 0.0, (double)nblocks * NT * (NT + 1) / 2 * 256 * 8.0);
    hipLaunchKernelGGL(gram_small_reduce_kernel, dim3(NT * (NT + 1) / 2), dim3(256), 0, st, Gpart, nblocks, NT, K, G);
  }
}

size_t launch_gram(hipStream_t st, const double* A, int64_t ldA, int64_t N, int64_t K, double* Gpart,
                   double* G, int num_cu, Ctx* prof) {
  const int64_t nslab = (N + GR - 1) / GR;
  if (K <= 128) {
    const int nt = (int)((K + 15) / 16);
    int64_t nblocks = (int64_t)num_cu * 2;  // two 8-wave workgroups per CU
    if (nblocks > nslab) nblocks = nslab;
    const size_t need = (size_t)nblocks * (nt * (nt + 1) / 2) * 256 * sizeof(double);
    if (Gpart == nullptr) return need;
    switch (nt) {
      case 1: launch_gram_small<1>(st, A, ldA, N, (int)K, Gpart, G, (int)nblocks, prof); break;
      case 2: launch_gram_small<2>(st, A, ldA, N, (int)K, Gpart, G, (int)nblocks, prof); break;
      case 3: launch_gram_small<3>(st, A, ldA, N, (int)K, Gpart, G, (int)nblocks, prof); break;
      case 4: launch_gram_small<4>(st, A, ldA, N, (int)K, Gpart, G, (int)nblocks, prof); break;
      case 5: launch_gram_small<5>(st, A, ldA, N, (int)K, Gpart, G, (int)nblocks, prof); break;
      case 6: launch_gram_small<6>(st, A, ldA, N, (int)K, Gpart, G, (int)nblocks, prof); break;
      case 7: launch_gram_small<7>(st, A, ldA, N, (int)K, Gpart, G, (int)nblocks, prof); break;
      default: launch_gram_small<8>(st, A, ldA, N, (int)K, Gpart, G, (int)nblocks, prof); break;
    }
    return need;
  }
  const int npanels = (int)((K + GP - 1) / GP);
  const int npairs = npanels * (npanels + 1) / 2;
  int64_t nsplit = ((int64_t)num_cu * 2 + npairs - 1) / npairs;  // ~2 workgroups per CU
  if (nsplit > nslab) nsplit = nslab;
  if (nsplit < 1) nsplit = 1;
  const size_t need = (size_t)nsplit * npairs * GP * GP * sizeof(double);
  if (Gpart == nullptr) return need;
  constexpr size_t lds = 2 * GP * GRP * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gram_pair_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  {
    ProfScope ps(prof, SI_K_GRAM, (double)N * (double)K * (double)(K + 1), (double)N * (double)K * 8.0);
    hipLaunchKernelGGL(gram_pair_kernel, dim3(npairs, (unsigned)nsplit), dim3(256), lds, st, A, ldA, N, (int)K,
                       npanels, Gpart);
  }
  {
    ProfScope ps(prof, SI_K_GRAM_RED, 0.0, (double)need);
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(npairs, GP * GP / 256), dim3(256), 0, st, Gpart, (int)nsplit, npanels, (int)K, G);
  }
  return need;
}

// ------------------------------------------------------------------------------------------------
// K3.  P = A * V_M  (N x K times K x M).  HBM-bound (intensity K*M/(4(K+M)) flop/B), so plain fp64 VALU
// FMAs: each thread owns two consecutive rows (16-B loads, lanes = consecutive rows => 1 KiB contiguous per
// wave per column of A) and MT accumulators per row; V[k][m] is wave-uniform and comes through scalar loads.
// Algorithmic bytes: N*(K+M)*8.
// ------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void project_kernel(const double* __restrict__ A, int64_t ldA, int64_t N,
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
      const double2 av = *reinterpret_cast<const double2*>(A + r + (int64_t)k * ldA);
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

int project_mpad(int M) {
  int m0 = 0;
  while (m0 < M) {
    const int rem = M - m0;
    m0 += rem > 24 ? 32 : rem > 16 ? 24 : rem > 8 ? 16 : 8;
  }
  return m0;
}

void launch_project(hipStream_t st, const double* A, int64_t ldA, int64_t N, int64_t K, const double* V,
                    int32_t M, int32_t Mpad, double* P, int64_t ldP, int num_cu) {
  int64_t blocks = (((N + 1) >> 1) + 255) / 256;
  if (blocks > (int64_t)num_cu * 8) blocks = (int64_t)num_cu * 8;
  if (blocks < 1) blocks = 1;
  int m0 = 0;
  while (m0 < M) {
    const int rem = M - m0;
    if (rem > 24) {
      hipLaunchKernelGGL((project_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP);
      m0 += 32;
    } else if (rem > 16) {
      hipLaunchKernelGGL((project_kernel<24>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP);
      m0 += 24;
    } else if (rem > 8) {
      hipLaunchKernelGGL((project_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP);
      m0 += 16;
    } else {
      hipLaunchKernelGGL((project_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP);
      m0 += 8;
    }
  }
}

}  // namespace si
