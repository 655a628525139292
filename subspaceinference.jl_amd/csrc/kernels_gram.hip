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

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a = (wave + pair + blockIdx.y) & 3;
  const int q = lane >> 4, c = lane & 15;

  d4 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = (d4){0.0, 0.0, 0.0, 0.0};

  const int64_t nslab = (N + GR - 1) / GR;
  // staging map: a wave-instruction covers 2 columns x 64 rows with 16 B per lane
  const int srow = (lane & 31) * 2;        // even row inside the slab
  const int scol0 = wave * 2 + (lane >> 5);  // column inside the panel, step 8 per pass

  for (int64_t slab = blockIdx.y; slab < nslab; slab += gridDim.y) {
    const int64_t r0 = slab * GR;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half == 1 && diag) break;
      const int panel = half == 0 ? I : J;
      double* dst = half == 0 ? sI : sJ;
#pragma unroll
      for (int pass = 0; pass < GP / 8; ++pass) {
        const int col = scol0 + pass * 8;
        const int gcol = panel * GP + col;
        const int64_t gr = r0 + srow;
        double2 v = make_double2(0.0, 0.0);
        if (gcol < K) {
          const double* src = A + gr + (int64_t)gcol * ldA;
          if (gr + 1 < N) {
            v = *reinterpret_cast<const double2*>(src);
          } else if (gr < N) {
            v.x = src[0];
          }
        }
        *reinterpret_cast<double2*>(dst + col * GRP + srow) = v;
      }
    }
    __syncthreads();
    const double* pa = sI + (a * 16 + c) * GRP + q;
    const double* pb = sJ + c * GRP + q;
#pragma unroll 4
    for (int s = 0; s < GR / 4; ++s) {
      const double fa = pa[4 * s];
      double fb[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = pb[b * 16 * GRP + 4 * s];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (!diag || b >= a) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[b], acc[b], 0, 0, 0);
      }
    }
    __syncthreads();
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

__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ Gpart, int nsplit,
                                                          int npanels, int K, double* __restrict__ G) {
  const int pair = blockIdx.x;
  const int npairs = gridDim.x;
  int I, J;
  pair_from_index(pair, npanels, I, J);
  for (int e = threadIdx.x; e < GP * GP; e += 256) {
    const int ii = e % GP, jj = e / GP;
    if (I == J && (ii >> 4) > (jj >> 4)) continue;  // tile below the diagonal: never computed
    const int gi = I * GP + ii, gj = J * GP + jj;
    if (gi >= K || gj >= K) continue;
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += Gpart[((int64_t)sp * npairs + pair) * (GP * GP) + e];
    if (I == J && (ii >> 4) == (jj >> 4)) {
      // diagonal tile: fully computed, both (ii,jj) and (jj,ii) are visited -- write own entry only
      G[gi + (int64_t)K * gj] = s;
    } else {
      G[gi + (int64_t)K * gj] = s;
      G[gj + (int64_t)K * gi] = s;
    }
  }
}

size_t launch_gram(hipStream_t st, const double* A, int64_t ldA, int64_t N, int64_t K, double* Gpart,
                   double* G, int num_cu, Ctx* prof) {
  const int npanels = (int)((K + GP - 1) / GP);
  const int npairs = npanels * (npanels + 1) / 2;
  const int64_t nslab = (N + GR - 1) / GR;
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
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(npairs), dim3(256), 0, st, Gpart, (int)nsplit, npanels, (int)K, G);
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
