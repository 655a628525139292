// K3 for wide subspaces (32 < M, K <= 128; cfg5: K = 128, M = 64):  P[N x M] = A[N x K] * V[K x M]  as a slab-streaming
// kernel on the fp64 matrix cores (round 2).  Replaces reference src/subspace_construction.jl:65 together with H1.
//
// The generic GEMM used before (launch_project_mfma -> gemm_f64_kernel<128, 64>) gives every workgroup 128 rows and a
// k loop of only 8 tiles: prologue and epilogue dominate (cfg5: 21 ms = 3.7 TB/s, 0.51 of either roofline).  Here the
// problem is treated as what it is -- a stream over A with a small resident right factor:
//   * one 4-wave workgroup per CU walks 32-row slabs of A; slabs arrive by LDS-DMA into a ring of NB buffers (the
//     staging of kernels_gram_wave.hip: unpadded [column][32 rows] image, counted vmcnt + raw barrier);
//   * wave w owns the 16 columns 16w .. 16w+15 of a 64-column panel of P for the whole kernel, so its V operand
//     (K x 16) is LOOP-INVARIANT: K/4 doubles per lane, loaded once into registers -- V never touches LDS;
//   * per k step a wave reads TWO A fragments (the two 16-row tiles of the slab) for two MFMAs; the LDS image is made
//     conflict-free for that access (lanes = consecutive rows of one column, q = 4 neighbouring columns) by permuting the
//     SOURCE rows: slot j of column c holds row pair j ^ (8 * (c & 1));
//   * the accumulators of a slab (2 tiles) are stored straight from registers: D[i = column of P][j = row], lanes along
//     the rows => 128-byte segments per store instruction.
// Algorithmic bytes N*(K+M)*8; at cfg5 (78.5 GB on one GPU) the kernel is HBM-bound: MFMA time per slab 8*NT*64 cycles.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "si_internal.h"

namespace si {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
constexpr int PR = 32;        // slab rows
constexpr int PJ_WAVES = 4;

template <int N, int I = 0, class F>
__device__ __forceinline__ void pj_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pj_static_for<N, I + 1>(f);
  }
}

// NT = ceil(K / 16) column tiles of A; NB ring buffers
// (the body lives in a __device__ function: device builtins inside lambdas of a __global__ template do not survive the
// host-side instantiation of its stub)
template <int NT, int NB>
__device__ __forceinline__ void project_glds_body(const double* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                  const double* __restrict__ V, int Mpad, int m0, int M, double* __restrict__ P,
                                                  int64_t ldP, double* sA) {
  constexpr int BUF = NT * 16 * PR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c = lane & 15;
  // LDS-DMA piece p: the 4 columns 16p + 4*wave .. +3; lane -> (column u = lane >> 4, slot j = lane & 15)
  const double* src[NT];
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    const int col = 16 * p + 4 * wave + (lane >> 4);
    const int rp = (lane & 15) ^ (8 * (col & 1));
    src[p] = A + (int64_t)(col < K ? col : K - 1) * ldA + 2 * rp;   // columns past K meet zero rows of V
  }
  auto issue_one = [&](int64_t roff, double* dst, auto PC) {
    constexpr int p = decltype(PC)::value;
    __builtin_amdgcn_global_load_lds(src[p] + roff, (lds_void_ptr)(dst + 16 * p * PR), 16, 0, 0);
  };
  auto issue = [&](int64_t slab, int buf) {
    const int64_t roff = slab * PR;
    double* dst = sA + buf * BUF + (4 * wave) * PR;
    pj_static_for<NT>([&](auto PC) { issue_one(roff, dst, PC); });
  };
  // this wave's right factor: V[k = 4s + q][m = m0 + 16*wave + c], zero outside K x M
  const int mcol = m0 + 16 * wave + c;
  double vf[4 * NT];
#pragma unroll
  for (int s = 0; s < 4 * NT; ++s) {
    const int k = 4 * s + q;
    vf[s] = (k < K && mcol < M) ? V[(int64_t)k * Mpad + mcol] : 0.0;
  }
  // A operand of row tile rt, k step s: element (column 4s + q, row 16 rt + c) -> slot ((8 rt + (c >> 1)) ^ (8 (q & 1)))
  int fa[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) fa[rt] = q * PR + 2 * ((8 * rt + (c >> 1)) ^ (8 * (q & 1))) + (c & 1);
  const int64_t nslab = (N + PR - 1) / PR;
  const int64_t stride = gridDim.x;
  int64_t slab = blockIdx.x;
  if (slab >= nslab) return;
  const int64_t last = nslab - 1;
  auto clamp = [&](int64_t sl) { return sl < last ? sl : last; };
#pragma unroll
  for (int b = 0; b < NB; ++b) issue(clamp(slab + b * stride), b);
  int ring = 0;
  const bool store_cols = 16 * wave < M - m0;   // a wave past the last column tile of the panel only stages
  // With one wave per SIMD nothing hides an instruction that is not an MFMA, so the memory traffic of a slab is dealt out
  // between the MFMAs of the NEXT slab: ONE barrier per slab (after it every wave's pieces of slab n have landed and every
  // wave is done reading slab n-1), then during the k steps of slab n: the NT LDS-DMA pieces that refill the buffer of
  // slab n-1 with slab n-1+NB (one behind every 4th step) and the 8 stores of slab n-1's accumulators (kept in registers).
  d4 accp[2];
  int64_t prow0 = 0;
  auto one_slab = [&](auto FIRST) {
    constexpr bool first = decltype(FIRST)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NB - 2) * NT) : "memory");   // conservative: counts stores as loads
    __builtin_amdgcn_s_barrier();
    const double* cur = sA + ring * BUF;
    const int prv = ring == 0 ? NB - 1 : ring - 1;
    const int64_t roff = clamp(slab + (int64_t)(NB - 1) * stride) * PR;
    double* pdst = sA + prv * BUF + (4 * wave) * PR;
    d4 acc[2];
    pj_static_for<4 * NT>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      const double a0 = cur[fa[0] + 4 * s * PR];
      const double a1 = cur[fa[1] + 4 * s * PR];
      if constexpr (s == 0) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[s], a0, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[s], a1, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[s], a0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[s], a1, acc[1], 0, 0, 0);
      }
      if constexpr (!first) {
        if constexpr (s % 4 == 1) issue_one(roff, pdst, std::integral_constant<int, s / 4>{});
        if constexpr (s % 4 == 3 && s / 4 < 8) {
          constexpr int j = s / 4, rt = j / 4, r = j % 4;
          const int m = m0 + 16 * wave + q + 4 * r;
          if (store_cols && m < M) P[prow0 + 16 * rt + c + (int64_t)m * ldP] = accp[rt][r];
        }
      }
    });
    if constexpr (!first && NT < 8) {   // stores that found no k step to hide behind (K < 128)
      pj_static_for<8 - NT>([&](auto JC) {
        constexpr int j = NT + decltype(JC)::value, rt = j / 4, r = j % 4;
        const int m = m0 + 16 * wave + q + 4 * r;
        if (store_cols && m < M) P[prow0 + 16 * rt + c + (int64_t)m * ldP] = accp[rt][r];
      });
    }
    accp[0] = acc[0];
    accp[1] = acc[1];
    prow0 = slab * PR;
    ring = ring + 1 == NB ? 0 : ring + 1;
    slab += stride;
  };
  one_slab(std::true_type{});
  while (slab < nslab) one_slab(std::false_type{});
  if (store_cols) {   // the last slab's accumulators
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wave + q + 4 * r;
        if (m < M) P[prow0 + 16 * rt + c + (int64_t)m * ldP] = accp[rt][r];
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail fetches must land before the workgroup retires
}

template <int NT, int NB, int OCC>
__global__ __launch_bounds__(64 * PJ_WAVES, OCC) void project_glds_kernel(const double* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                                        const double* __restrict__ V, int Mpad, int m0, int M,
                                                                        double* __restrict__ P, int64_t ldP) {
  extern __shared__ double sA[];  // [NB][NT*16][32]
  project_glds_body<NT, NB>(A, ldA, N, K, V, Mpad, m0, M, P, ldP, sA);
}

// Two workgroups per CU with two ring buffers each is the shipped configuration (cfg5 share: 1.92 ms against 2.20 ms for
// one workgroup per CU with four buffers: one workgroup's barrier / issue gaps are the other's MFMA time);
// the development build's SI_PROJECT_OCC1=1 selects the other one for comparison runs
#ifdef SI_DEV_KNOBS
static bool project_two_per_cu() {
  static const bool v = [] {
    const char* e = getenv("SI_PROJECT_OCC1");
    return !(e && e[0] == '1');
  }();
  return v;
}
#else
static constexpr bool project_two_per_cu() { return true; }
#endif

template <int NT>
static void launch_project_nt(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, const double* V, int Mpad, int m0,
                              int M, double* P, int64_t ldP, int num_cu) {
  const int64_t nslab = (N + PR - 1) / PR;
  if (project_two_per_cu()) {
    // two workgroups per CU, two ring buffers each: one workgroup's barrier / issue gaps are the other's MFMA time
    constexpr int NB = 2;
    constexpr size_t lds = (size_t)NB * NT * 16 * PR * sizeof(double);
    static LdsOptIn optin;
    optin.ensure(reinterpret_cast<const void*>(project_glds_kernel<NT, NB, 2>), lds);
    const int64_t grid = std::min<int64_t>((int64_t)num_cu * 2, nslab);
    hipLaunchKernelGGL((project_glds_kernel<NT, NB, 2>), dim3((unsigned)grid), dim3(64 * PJ_WAVES), lds, st, A, ldA, N, K, V, Mpad,
                       m0, M, P, ldP);
    return;
  }
  constexpr int NB = (150 * 1024) / (NT * 16 * PR * 8) >= 4 ? 4 : (150 * 1024) / (NT * 16 * PR * 8) >= 3 ? 3 : 2;
  constexpr size_t lds = (size_t)NB * NT * 16 * PR * sizeof(double);
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(project_glds_kernel<NT, NB, 1>), lds);
  const int64_t grid = std::min<int64_t>(num_cu, nslab);
  hipLaunchKernelGGL((project_glds_kernel<NT, NB, 1>), dim3((unsigned)grid), dim3(64 * PJ_WAVES), lds, st, A, ldA, N, K, V, Mpad, m0,
                     M, P, ldP);
}

// ---- the same stream for an fp32-STORED deviation matrix (si_construct_set_storage(SI_F32), round 4): half the bytes per row
// of A.  The slab image is [column][32 rows] of floats (128 bytes per column: one LDS-DMA instruction moves 8 columns), the
// operand read is a ds_read_b32 widened to fp64 on the way into the MFMA (2 conversions per 2 MFMAs), V and the accumulators
// are fp64 as before.  Conflict-free reads: lanes q = 0..3 read columns 4s + q, 128 bytes = 32 banks apart, so columns whose
// bit 1 is set store their two 16-row halves exchanged (done on the DMA source: slot j of column c holds row quad
// j ^ (4 * ((c >> 1) & 1))).  Two workgroups per CU, two buffers each.
template <int NT>
__device__ __forceinline__ void project_glds_f32_body(const float* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                      const double* __restrict__ V, int Mpad, int m0, int M, double* __restrict__ P,
                                                      int64_t ldP, float* sA) {
  constexpr int NB = 2;
  constexpr int BUF = NT * 16 * PR;              // floats per buffer
  constexpr int NI = 2 * NT;                     // DMA instructions (8 columns each) per slab
  constexpr int NP = (NI + PJ_WAVES - 1) / PJ_WAVES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c = lane & 15;
  const float* src[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int i = p * PJ_WAVES + wave;           // instruction i: columns 8 i .. 8 i + 7
    const int col = 8 * (i < NI ? i : 0) + (lane >> 3);
    const int jq = (lane & 7) ^ (4 * ((col >> 1) & 1));
    src[p] = A + (int64_t)(col < K ? col : K - 1) * ldA + 4 * jq;   // columns past K meet zero rows of V
  }
  auto issue_one = [&](int64_t roff, float* dst, auto PC) {
    constexpr int p = decltype(PC)::value;
    if (p * PJ_WAVES + wave < NI)
      __builtin_amdgcn_global_load_lds(src[p] + roff, (lds_void_ptr)(dst + (p * PJ_WAVES + wave) * 8 * PR), 16, 0, 0);
  };
  auto issue = [&](int64_t slab, int buf) {
    const int64_t roff = slab * PR;
    float* dst = sA + buf * BUF;
    pj_static_for<NP>([&](auto PC) { issue_one(roff, dst, PC); });
  };
  const int mcol = m0 + 16 * wave + c;
  double vf[4 * NT];
#pragma unroll
  for (int s = 0; s < 4 * NT; ++s) {
    const int k = 4 * s + q;
    vf[s] = (k < K && mcol < M) ? V[(int64_t)k * Mpad + mcol] : 0.0;
  }
  int fa[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) fa[rt] = q * PR + ((16 * rt + c) ^ (16 * ((q >> 1) & 1)));
  const int64_t nslab = (N + PR - 1) / PR;
  const int64_t stride = gridDim.x;
  int64_t slab = blockIdx.x;
  if (slab >= nslab) return;
  const int64_t last = nslab - 1;
  auto clamp = [&](int64_t sl) { return sl < last ? sl : last; };
#pragma unroll
  for (int b = 0; b < NB; ++b) issue(clamp(slab + b * stride), b);
  int ring = 0;
  const bool store_cols = 16 * wave < M - m0;
  d4 accp[2];
  int64_t prow0 = 0;
  auto one_slab = [&](auto FIRST) {
    constexpr bool first = decltype(FIRST)::value;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const float* cur = sA + ring * BUF;
    const int prv = ring ^ 1;
    const int64_t roff = clamp(slab + stride) * PR;
    float* pdst = sA + prv * BUF;
    d4 acc[2];
    float an0[4 * NT + 2], an1[4 * NT + 2];   // operand reads two k steps ahead of their MFMAs (measured: 2 % on this kernel)
    an0[0] = cur[fa[0]];
    an1[0] = cur[fa[1]];
    an0[1] = cur[fa[0] + 4 * PR];
    an1[1] = cur[fa[1] + 4 * PR];
    pj_static_for<4 * NT>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      if constexpr (s + 2 < 4 * NT) {
        an0[s + 2] = cur[fa[0] + 4 * (s + 2) * PR];
        an1[s + 2] = cur[fa[1] + 4 * (s + 2) * PR];
      }
      const double a0 = (double)an0[s];
      const double a1 = (double)an1[s];
      if constexpr (s == 0) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[s], a0, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[s], a1, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
      } else {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[s], a0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(vf[s], a1, acc[1], 0, 0, 0);
      }
      if constexpr (!first) {
        // the refill of the buffer slab n-1 has left (NP pieces, one behind every 8th k step) and the 8 stores of its accumulators
        if constexpr (s % 8 == 1 && s / 8 < NP) issue_one(roff, pdst, std::integral_constant<int, s / 8>{});
        if constexpr (s % 4 == 3 && s / 4 < 8) {
          constexpr int j = s / 4, rt = j / 4, r = j % 4;
          const int m = m0 + 16 * wave + q + 4 * r;
          if (store_cols && m < M) P[prow0 + 16 * rt + c + (int64_t)m * ldP] = accp[rt][r];
        }
      }
    });
    if constexpr (!first) {
      // pieces and stores that found no k step to hide behind (short K)
      pj_static_for<NP>([&](auto PC) {
        constexpr int p = decltype(PC)::value;
        if constexpr (8 * p + 1 >= 4 * NT) issue_one(roff, pdst, PC);
      });
      if constexpr (NT < 8) {
        pj_static_for<8 - NT>([&](auto JC) {
          constexpr int j = NT + decltype(JC)::value, rt = j / 4, r = j % 4;
          const int m = m0 + 16 * wave + q + 4 * r;
          if (store_cols && m < M) P[prow0 + 16 * rt + c + (int64_t)m * ldP] = accp[rt][r];
        });
      }
    }
    accp[0] = acc[0];
    accp[1] = acc[1];
    prow0 = slab * PR;
    ring ^= 1;
    slab += stride;
  };
  one_slab(std::true_type{});
  while (slab < nslab) one_slab(std::false_type{});
  if (store_cols) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wave + q + 4 * r;
        if (m < M) P[prow0 + 16 * rt + c + (int64_t)m * ldP] = accp[rt][r];
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NT>
__global__ __launch_bounds__(64 * PJ_WAVES, 2) void project_glds_f32_kernel(const float* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                                          const double* __restrict__ V, int Mpad, int m0, int M,
                                                                          double* __restrict__ P, int64_t ldP) {
  extern __shared__ float sA32[];  // [2][NT*16][32]
  project_glds_f32_body<NT>(A, ldA, N, K, V, Mpad, m0, M, P, ldP, sA32);
}

template <int NT>
static void launch_project_f32_nt(hipStream_t st, const float* A, int64_t ldA, int64_t N, int K, const double* V, int Mpad, int m0,
                                  int M, double* P, int64_t ldP, int num_cu) {
  const int64_t nslab = (N + PR - 1) / PR;
  constexpr size_t lds = (size_t)2 * NT * 16 * PR * sizeof(float);
  const int64_t grid = std::min<int64_t>((int64_t)num_cu * 2, nslab);
  hipLaunchKernelGGL((project_glds_f32_kernel<NT>), dim3((unsigned)grid), dim3(64 * PJ_WAVES), lds, st, A, ldA, N, K, V, Mpad, m0, M, P,
                     ldP);
}

// fp32-stored A: P[:, m] for m < M in 64-column panels; false when the shape is not covered (K > 128)
bool launch_project_stream_f32(hipStream_t st, const float* A, int64_t ldA, int64_t N, int64_t K, const double* V, int32_t M,
                               int32_t Mpad, double* P, int64_t ldP, int num_cu) {
  if (K > 128 || K <= 0) return false;
  for (int m0 = 0; m0 < M; m0 += 64) {
    switch ((int)(K + 15) / 16) {
      case 1: launch_project_f32_nt<1>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 2: launch_project_f32_nt<2>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 3: launch_project_f32_nt<3>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 4: launch_project_f32_nt<4>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 5: launch_project_f32_nt<5>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 6: launch_project_f32_nt<6>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 7: launch_project_f32_nt<7>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      default: launch_project_f32_nt<8>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
    }
  }
  return true;
}

// P[:, m] for m < M in 64-column panels; false when the shape is not covered (K > 128): the caller falls back to the GEMM
bool launch_project_stream(hipStream_t st, const double* A, int64_t ldA, int64_t N, int64_t K, const double* V, int32_t M,
                           int32_t Mpad, double* P, int64_t ldP, int num_cu) {
  if (K > 128 || K <= 0) return false;
  for (int m0 = 0; m0 < M; m0 += 64) {
    switch ((int)(K + 15) / 16) {
      case 1: launch_project_nt<1>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 2: launch_project_nt<2>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 3: launch_project_nt<3>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 4: launch_project_nt<4>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 5: launch_project_nt<5>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 6: launch_project_nt<6>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      case 7: launch_project_nt<7>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
      default: launch_project_nt<8>(st, A, ldA, N, (int)K, V, Mpad, m0, M, P, ldP, num_cu); break;
    }
  }
  return true;
}

}  // namespace si
