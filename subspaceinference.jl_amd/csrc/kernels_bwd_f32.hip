// The reverse sweep of a Dense chain in fp32 (round 5): the training step of reference src/subspace_construction.jl:39-43 in the
// CALLER's precision.  With a Float32 Flux model and Float32 data the reference's Zygote pass is Float32 throughout (sgemm
// forward and reverse, Float32 loss); rounds 1-4 promoted such data to Float64 and trained in fp64 -- wider than the reference
// and twice the matrix time.  Forward: kernels_gemm_f32.hip (v_mfma_f32_32x32x2_f32).  Here:
//   dW_l = Delta_l * H_{l-1}'        dw_f32_dma_kernel: split-K over the batch, both operands "row-fast" (one observation = R
//                                    contiguous floats), global -> LDS by LDS-DMA, ds_read_b32 fragments straight off the
//                                    linear [k][R] image (32 consecutive rows: conflict-free), partials summed in fixed order
//   Delta_{l-1} = (W_l' Delta_l) .* act'(H_{l-1})   the forward kernel on W_l' (transposed once per step: 4 MB) + one
//                                    elementwise pass that also gives db_{l-1} = rowsum(Delta_{l-1})
//   narrow head (out <= 4)           one pass over H_{L-1} as in the fp64 sweep (tail_bwd_kernel), fp32 in / out, fp64 sums
// Every sum over the batch is accumulated in a fixed order (bit-reproducible from run to run); sums of more than a tile are kept
// in fp64 and rounded once to the Float32 gradient -- at least as accurate as the reference's sgemm, not bit-equal to it (BLAS
// blocks its sums differently): tests/test_gpu_train_f32.py states the measured tolerance against the oracle's Float32 path.
#include <algorithm>
#include <type_traits>

#include "kernels_gemm.h"

namespace si {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr_f;

template <int N, int I = 0, class F>
__device__ __forceinline__ void bf_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    bf_static_for<N, I + 1>(f);
  }
}

// ---- dW partials: part[split][m + Mrows * n] = sum_{k in split} A[m + lda k] * Bm[n + ldb k] -------------------------------
// 128 x 128 output tile per 256-thread workgroup (waves 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles of 32 x 32: 64 accumulator
// registers), k tiles of 16 observations in two LDS stages: s_waitcnt vmcnt(0); s_barrier; DMA of tile kt+1; 32 MFMAs on tile kt.
// Needs Mrows % 4 == 0, Ncols % 4 == 0, whole k tiles (Kdim % 16 == 0) and 16-byte aligned operands (dw_f32_ok).
template <int BM, int BN>
__device__ __forceinline__ void dw_f32_dma_body(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
                                                float* __restrict__ C, int Mrows, int Ncols, int64_t Kdim, int64_t ksplit, int nMt,
                                                int nNt, int nsplit, float* smem) {
  constexpr int NWAVES = 4, WM = 2, WN = 2;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int STAGE = 16 * (BM + BN);                       // floats per stage
  constexpr int NA = 16 * BM / 256, NBI = 16 * BN / 256;      // 1 KiB DMA instructions per k tile: A part, B part
  static_assert(NA % NWAVES == 0 && NBI % NWAVES == 0, "whole DMA slots per wave");
  constexpr int SA_N = NA / NWAVES, SB_N = NBI / NWAVES;
  // workgroup -> (tile, split): one round of workgroups, the tiles of one k range on one XCD (they walk the same columns of
  // Delta and H and find them in that L2), as in dw_f64_dma_kernel
  const int tiles = nMt * nNt;
  const int total = tiles * nsplit;
  const int per_xcd = (total + 7) >> 3;
  const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || work >= total) return;   // uniform per workgroup, before any barrier
  const int64_t split = work / tiles;
  const int tile = work % tiles;
  const int mt = tile % nMt, nt = tile / nMt;
  const int64_t k0 = split * ksplit;
  int64_t klen = Kdim - k0;
  if (klen > ksplit) klen = ksplit;
  if (klen <= 0) return;
  const int nk = (int)(klen / 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int m0 = mt * BM, n0 = nt * BN;

  // DMA plan: slot s of an operand is instruction wave + 4 s; a lane keeps one 32-bit BYTE offset per slot (its 16 bytes =
  // four consecutive rows of one observation), the tile's base address is wave-uniform
  uint32_t aoff[SA_N], boff[SB_N];
#pragma unroll
  for (int s = 0; s < SA_N; ++s) {
    const int d = (wave + NWAVES * s) * 64 + lane;   // 16-byte chunk inside the A image: row k = d / (BM/4)
    const int k = d / (BM / 4), pp = d % (BM / 4);
    int m = m0 + 4 * pp;
    if (m > Mrows - 4) m = Mrows - 4;                 // clamped rows only feed outputs that are never stored
    aoff[s] = (uint32_t)(m + (int)lda * k) * 4u;
  }
#pragma unroll
  for (int s = 0; s < SB_N; ++s) {
    const int d = (wave + NWAVES * s) * 64 + lane;
    const int k = d / (BN / 4), pp = d % (BN / 4);
    int n = n0 + 4 * pp;
    if (n > Ncols - 4) n = Ncols - 4;
    boff[s] = (uint32_t)(n + (int)ldb * k) * 4u;
  }
  auto issue = [&](int kt, int buf) {
    float* dst = smem + buf * STAGE;
    const char* ta = reinterpret_cast<const char*>(A + lda * (k0 + 16 * (int64_t)kt));
    const char* tb = reinterpret_cast<const char*>(Bm + ldb * (k0 + 16 * (int64_t)kt));
    bf_static_for<SA_N>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(ta + aoff[s]), (lds_void_ptr_f)(dst + (wave + NWAVES * s) * 256), 16, 0, 0);
    });
    bf_static_for<SB_N>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(tb + boff[s]),
                                       (lds_void_ptr_f)(dst + 16 * BM + (wave + NWAVES * s) * 256), 16, 0, 0);
    });
  };

  f16v acc[TN][TM];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][a][r] = 0.0f;

  // fragment offsets (floats inside a stage) of k step 0 (two observations per MFMA: lane l holds row l % 32 of observation l / 32)
  const int half = lane >> 5, l32 = lane & 31;
  int offA[TM], offB[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) offA[a] = half * BM + (wm * TM + a) * 32 + l32;
#pragma unroll
  for (int b = 0; b < TN; ++b) offB[b] = 16 * BM + half * BN + (wn * TN + b) * 32 + l32;

  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's pieces of tile kt are in; every wave is done reading tile kt-1
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const float* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = st[offA[a] + 2 * s * BM];
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = st[offB[b] + 2 * s * BN];
      // A operand = the H rows (n), B operand = the Delta rows (m): D[n][m], lane l holds column m = l % 32 -- consecutive lanes,
      // consecutive elements of a column of dW
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int a = 0; a < TM; ++a) acc[b][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[b], fa[a], acc[b][a], 0, 0, 0);
    }
  }
  float* Cout = C + split * (int64_t)Mrows * Ncols;
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int m = m0 + (wm * TM + a) * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wn * TN + b) * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        if (m < Mrows && n < Ncols) Cout[m + (int64_t)Mrows * n] = acc[b][a][r];
      }
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void dw_f32_dma_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                            int64_t ldb, float* __restrict__ C, int Mrows, int Ncols, int64_t Kdim,
                                                            int64_t ksplit, int nMt, int nNt, int nsplit) {
  extern __shared__ float smem_dwf[];
  dw_f32_dma_body<BM, BN>(A, lda, Bm, ldb, C, Mrows, Ncols, Kdim, ksplit, nMt, nNt, nsplit, smem_dwf);
}

// any shape / alignment: one thread per element of a split's partial, the batch walked in order (small or odd layers only)
__global__ __launch_bounds__(256) void dw_f32_generic_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                             int64_t ldb, float* __restrict__ C, int Mrows, int Ncols, int64_t Kdim,
                                                             int64_t ksplit) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)Mrows * Ncols) return;
  const int m = (int)(e % Mrows), n = (int)(e / Mrows);
  const int64_t k0 = (int64_t)blockIdx.y * ksplit;
  int64_t k1 = k0 + ksplit;
  if (k1 > Kdim) k1 = Kdim;
  float s = 0.0f;
  for (int64_t k = k0; k < k1; ++k) s = fmaf(A[m + lda * k], Bm[n + ldb * k], s);
  C[(int64_t)blockIdx.y * Mrows * Ncols + e] = s;
}

// dst[e] = (float) sum_split part[split][e] (fp64 accumulation in split order: bit-reproducible)
__global__ __launch_bounds__(256) void split_reduce_f32_kernel(const float* __restrict__ part, int nsplit, int64_t elems,
                                                               float* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < elems; e += stride) {
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += (double)part[(int64_t)sp * elems + e];
    dst[e] = (float)s;
  }
}

struct DwPlanF {
  bool dma;
  int nsplit;
  int64_t ks;
};
static DwPlanF plan_dw_f32(const float* Delta, const float* Hprev, const float* part, int32_t out, int32_t in, int64_t B, int num_cu) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  DwPlanF p;
  p.dma = B % 16 == 0 && B >= 16 && out % 4 == 0 && in % 4 == 0 && out >= 4 && in >= 4 && al(Delta) && al(Hprev) && al(part);
  const int64_t tiles = p.dma ? (int64_t)((out + 127) / 128) * ((in + 127) / 128) : ((int64_t)out * in + 255) / 256;
  const int64_t slots = (int64_t)num_cu * (p.dma ? 4 : 16), maxsplit = (B + 255) / 256;
  int64_t ns = std::max<int64_t>(1, std::min(slots / std::max<int64_t>(tiles, 1), maxsplit));
  p.ks = ((B + ns - 1) / ns + 15) / 16 * 16;
  p.nsplit = (int)((B + p.ks - 1) / p.ks);
  return p;
}
// floats of scratch launch_backward_weight_f32 needs for a layer at any batch of up to B columns
size_t backward_weight_f32_part_elems(int32_t out, int32_t in, int64_t B, int num_cu) {
  const int64_t maxsplit = (B + 255) / 256;
  const int64_t t_dma = (int64_t)((out + 127) / 128) * ((in + 127) / 128), t_gen = ((int64_t)out * in + 255) / 256;
  const int64_t worst = std::max<int64_t>(
      1, std::min(maxsplit, std::max((int64_t)num_cu * 4 / std::max<int64_t>(t_dma, 1), (int64_t)num_cu * 16 / std::max<int64_t>(t_gen, 1))));
  return (size_t)(worst + 1) * out * in;
}

void launch_backward_weight_f32(hipStream_t st, const float* Delta, const float* Hprev, float* part, int32_t out, int32_t in, int64_t B,
                                int num_cu, float* dW) {
  const DwPlanF p = plan_dw_f32(Delta, Hprev, part, out, in, B, num_cu);
  if (p.dma) {
    constexpr size_t lds = 2 * 16 * (128 + 128) * sizeof(float);
    const int nMt = (out + 127) / 128, nNt = (in + 127) / 128;
    const int64_t total = (int64_t)nMt * nNt * p.nsplit;
    hipLaunchKernelGGL((dw_f32_dma_kernel<128, 128>), dim3((unsigned)((total + 7) / 8 * 8)), dim3(256), lds, st, Delta, (int64_t)out, Hprev,
                       (int64_t)in, part, (int)out, (int)in, B, p.ks, nMt, nNt, p.nsplit);
  } else {
    const int64_t elems = (int64_t)out * in;
    hipLaunchKernelGGL(dw_f32_generic_kernel, dim3((unsigned)((elems + 255) / 256), (unsigned)p.nsplit), dim3(256), 0, st, Delta, (int64_t)out,
                       Hprev, (int64_t)in, part, (int)out, (int)in, B, p.ks);
  }
  const int64_t elems = (int64_t)out * in;
  int64_t blocks = (elems + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(split_reduce_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, p.nsplit, elems, dW);
}

// ---- small kernels ------------------------------------------------------------------------------------------------------
// Wt[k + in * i] = W[i + out * k]: the layer's weights transposed, so that W' Delta runs on the forward kernel
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ W, int out, int in, float* __restrict__ Wt) {
  __shared__ float tile[32][33];
  const int i0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int i = i0 + tx, k = k0 + j;
    tile[j][tx] = (i < out && k < in) ? W[i + (int64_t)out * k] : 0.0f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int k = k0 + tx, i = i0 + j;
    if (k < in && i < out) Wt[k + (int64_t)in * i] = tile[tx][j];
  }
}
void launch_transpose_f32(hipStream_t st, const float* W, int32_t out, int32_t in, float* Wt) {
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((out + 31) / 32, (in + 31) / 32), dim3(256), 0, st, W, (int)out, (int)in, Wt);
}


// D[i + rows b] = G[i + rows b] * act'(H[i + rows b]) (in place allowed) and part[chunk][i] = sum over the chunk's columns of D
// (fp64 partial sums, chunks summed in order by rowsum_final_f32_kernel: db of the layer below)
constexpr int RSF_CHUNKS = 64;
__global__ __launch_bounds__(256) void mul_dact_rowsum_f32_kernel(const float* __restrict__ G, const float* __restrict__ H, int rows,
                                                                  int64_t B, int act, float* __restrict__ D, double* __restrict__ part) {
  __shared__ double red[8][33];
  const int il = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il;
  const int64_t per = (B + RSF_CHUNKS - 1) / RSF_CHUNKS;
  const int64_t b0 = (int64_t)blockIdx.y * per;
  int64_t b1 = b0 + per;
  if (b1 > B) b1 = B;
  double s = 0.0;
  if (i < rows)
    for (int64_t b = b0 + bl; b < b1; b += 8) {
      const int64_t e = i + (int64_t)rows * b;
      const float d = H != nullptr ? G[e] * dact_f32(H[e], act) : G[e];
      if (D != nullptr) D[e] = d;
      s += (double)d;
    }
  red[bl][il] = s;
  __syncthreads();
  if (bl == 0 && i < rows) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][il];
    part[(int64_t)blockIdx.y * rows + i] = t;
  }
}
// db = rowsum(G) alone (the dX GEMM's epilogue has already applied act'): 16-byte loads, four features per thread, RS4_CHUNKS
// slices of the batch so that 2048 workgroups stream the panel
constexpr int RS4_CHUNKS = 256;
__global__ __launch_bounds__(256) void rowsum4_f32_kernel(const float* __restrict__ G, int rows, int64_t B, double* __restrict__ part) {
  __shared__ double red[8][32][4];
  const int il = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int i = (blockIdx.x * 32 + il) * 4;
  const int64_t per = (B + RS4_CHUNKS - 1) / RS4_CHUNKS;
  const int64_t b0 = (int64_t)blockIdx.y * per;
  int64_t b1 = b0 + per;
  if (b1 > B) b1 = B;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (i < rows) {
    int64_t b = b0 + bl;
    for (; b + 24 < b1; b += 32) {   // four loads in flight
      f4v v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f4v*>(G + i + (int64_t)rows * (b + 8 * u));
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += (double)v[u][j];
    }
    for (; b < b1; b += 8) {
      const f4v v = *reinterpret_cast<const f4v*>(G + i + (int64_t)rows * b);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += (double)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[bl][il][j] = s[j];
  __syncthreads();
  if (bl < 4 && i < rows) {   // thread (bl, il) finishes feature i + bl
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][il][bl];
    part[(int64_t)blockIdx.y * rows + i + bl] = t;
  }
}
// ... and of a NARROW panel (rows <= 8: the head's Delta): the threads of a workgroup run along the batch
__global__ __launch_bounds__(256) void rowsum_narrow_f32_kernel(const float* __restrict__ G, int rows, int64_t B, double* __restrict__ part) {
  __shared__ double red[4][8];
  const int64_t per = (B + RSF_CHUNKS - 1) / RSF_CHUNKS;
  const int64_t b0 = (int64_t)blockIdx.x * per;
  int64_t b1 = b0 + per;
  if (b1 > B) b1 = B;
  double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t b = b0 + threadIdx.x; b < b1; b += 256)
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (r < rows) s[r] += (double)G[r + (int64_t)rows * b];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s[r] += __shfl_down(s[r], off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][r] = s[r];
  }
  __syncthreads();
  if ((int)threadIdx.x < rows) part[(int64_t)blockIdx.x * rows + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// db[i] = (float) sum over chunks of part[chunk][i]: 32 rows per workgroup, eight threads per row each summing every eighth
// chunk (independent loads), the eight combined in order
__global__ __launch_bounds__(256) void rowsum_final_f32_kernel(const double* __restrict__ part, int rows, int chunks, float* __restrict__ db) {
  __shared__ double red[8][33];
  const int il = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il;
  double s = 0.0;
  if (i < rows)
    for (int ch = q; ch < chunks; ch += 8) s += part[(int64_t)ch * rows + i];
  red[q][il] = s;
  __syncthreads();
  if (q == 0 && i < rows) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][il];
    db[i] = (float)t;
  }
}
size_t rowsum_f32_part_elems(int max_rows) { return (size_t)RS4_CHUNKS * (size_t)max_rows; }
// D = G .* act'(H) (H == nullptr: D = G, nothing stored when D == nullptr either) and db = rowsum(D)
void launch_mul_dact_rowsum_f32(hipStream_t st, const float* G, const float* H, int rows, int64_t B, int act, float* D, double* part,
                                float* db) {
  if (H == nullptr && D == nullptr && rows % 4 == 0 && (reinterpret_cast<uintptr_t>(G) & 15u) == 0) {
    hipLaunchKernelGGL(rowsum4_f32_kernel, dim3((rows / 4 + 31) / 32, RS4_CHUNKS), dim3(256), 0, st, G, rows, B, part);
    hipLaunchKernelGGL(rowsum_final_f32_kernel, dim3((rows + 31) / 32), dim3(256), 0, st, part, rows, RS4_CHUNKS, db);
    return;
  }
  if (H == nullptr && D == nullptr && rows <= 8) {
    hipLaunchKernelGGL(rowsum_narrow_f32_kernel, dim3(RSF_CHUNKS), dim3(256), 0, st, G, rows, B, part);
    hipLaunchKernelGGL(rowsum_final_f32_kernel, dim3(1), dim3(256), 0, st, part, rows, RSF_CHUNKS, db);
    return;
  }
  hipLaunchKernelGGL(mul_dact_rowsum_f32_kernel, dim3((rows + 31) / 32, RSF_CHUNKS), dim3(256), 0, st, G, H, rows, B, act, D, part);
  hipLaunchKernelGGL(rowsum_final_f32_kernel, dim3((rows + 31) / 32), dim3(256), 0, st, part, rows, RSF_CHUNKS, db);
}

// Delta_L[e] = (float)(scale * (Y[e] - Yhat[e]) * act_L'(Yhat[e])): the seed of the reverse sweep; Yhat in fp64 (the fused
// head's output) or fp32 (a wide last layer)
template <typename YT>
__global__ __launch_bounds__(256) void delta_out_f32_kernel(const double* __restrict__ Y, const YT* __restrict__ Yhat, int64_t d,
                                                            double scale, int act, float* __restrict__ delta) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < d; e += stride) {
    const double yh = (double)Yhat[e];
    delta[e] = (float)(scale * (Y[e] - yh) * dact_full(yh, act));
  }
}
void launch_delta_out_f32(hipStream_t st, const double* Y, const double* Yhat64, const float* Yhat32, int64_t d, double scale, int act,
                          float* delta) {
  int64_t blocks = (d + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (Yhat64)
    hipLaunchKernelGGL(delta_out_f32_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, st, Y, Yhat64, d, scale, act, delta);
  else
    hipLaunchKernelGGL(delta_out_f32_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, Y, Yhat32, d, scale, act, delta);
}

// ---- reverse sweep through a NARROW last layer in one pass over its fp32 input H (F x B): fp32 in / out, the sums over the
// batch in fp64 chunk partials (tail_bwd_kernel of kernels_bwd.hip, same structure)
constexpr int TAILBF_CHUNKS = 512;
template <int OL>
__global__ __launch_bounds__(256) void tail_bwd_f32_kernel(const float* __restrict__ W, const float* __restrict__ Delta,
                                                           const float* __restrict__ H, int F, int64_t B, int act_prev,
                                                           float* __restrict__ DeltaPrev, double* __restrict__ part, int64_t cols) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= F) return;
  const int64_t b0 = (int64_t)blockIdx.y * cols;
  int64_t b1 = b0 + cols;
  if (b1 > B) b1 = B;
  float w[OL];
  double dw[OL], db = 0.0;
#pragma unroll
  for (int o = 0; o < OL; ++o) {
    w[o] = W[o + (int64_t)OL * i];
    dw[o] = 0.0;
  }
#pragma unroll 4
  for (int64_t b = b0; b < b1; ++b) {
    const float h = H[i + (int64_t)F * b];
    float t = 0.0f;
#pragma unroll
    for (int o = 0; o < OL; ++o) {
      const float dl = Delta[o + (int64_t)OL * b];
      t = fmaf(w[o], dl, t);
      dw[o] += (double)dl * (double)h;
    }
    const float d = t * dact_f32(h, act_prev);
    db += (double)d;
    DeltaPrev[i + (int64_t)F * b] = d;
  }
  double* pp = part + (int64_t)blockIdx.y * (OL + 1) * F;
#pragma unroll
  for (int o = 0; o < OL; ++o) pp[(int64_t)o * F + i] = dw[o];
  pp[(int64_t)OL * F + i] = db;
}
__global__ __launch_bounds__(256) void tail_bwd_reduce_f32_kernel(const double* __restrict__ part, int chunks, int OL, int F,
                                                                  float* __restrict__ dW, float* __restrict__ dbprev) {
  __shared__ double red[8][33];
  const int il = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int total = (OL + 1) * F;
  const int idx = blockIdx.x * 32 + il;
  double s = 0.0;
  if (idx < total)
    for (int c = cl; c < chunks; c += 8) s += part[(int64_t)c * total + idx];
  red[cl][il] = s;
  __syncthreads();
  if (cl == 0 && idx < total) {
    const double t = ((red[0][il] + red[1][il]) + (red[2][il] + red[3][il])) + ((red[4][il] + red[5][il]) + (red[6][il] + red[7][il]));
    const int o = idx / F, i = idx - o * F;
    if (o < OL)
      dW[o + (int64_t)OL * i] = (float)t;
    else
      dbprev[i] = (float)t;
  }
}
size_t tail_bwd_f32_part_elems(int32_t out_last, int32_t F) { return (size_t)TAILBF_CHUNKS * (out_last + 1) * F; }
void launch_tail_bwd_f32(hipStream_t st, const float* W, const float* Delta, const float* H, int32_t out_last, int32_t F, int64_t B,
                         int32_t act_prev, float* DeltaPrev, double* part, float* dW, float* dbprev) {
  int chunks = TAILBF_CHUNKS;
  if (chunks > B) chunks = (int)B;
  const int64_t cols = (B + chunks - 1) / chunks;
  chunks = (int)((B + cols - 1) / cols);
  const dim3 grid((F + 255) / 256, chunks);
  switch (out_last) {
    case 1: hipLaunchKernelGGL((tail_bwd_f32_kernel<1>), grid, dim3(256), 0, st, W, Delta, H, (int)F, B, (int)act_prev, DeltaPrev, part, cols); break;
    case 2: hipLaunchKernelGGL((tail_bwd_f32_kernel<2>), grid, dim3(256), 0, st, W, Delta, H, (int)F, B, (int)act_prev, DeltaPrev, part, cols); break;
    case 3: hipLaunchKernelGGL((tail_bwd_f32_kernel<3>), grid, dim3(256), 0, st, W, Delta, H, (int)F, B, (int)act_prev, DeltaPrev, part, cols); break;
    default: hipLaunchKernelGGL((tail_bwd_f32_kernel<4>), grid, dim3(256), 0, st, W, Delta, H, (int)F, B, (int)act_prev, DeltaPrev, part, cols); break;
  }
  const int total = (out_last + 1) * F;
  hipLaunchKernelGGL(tail_bwd_reduce_f32_kernel, dim3((total + 31) / 32), dim3(256), 0, st, part, chunks, (int)out_last, (int)F, dW, dbprev);
}

// dst[i] = (float) src[i]
__global__ __launch_bounds__(256) void narrow_copy_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}

}  // namespace si
