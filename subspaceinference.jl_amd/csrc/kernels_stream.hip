// HBM-bound streaming kernels of the hot path (gfx950): K1 swa_dev_push, K4 reconstruct, the SSE
// reduction of K5 and the tiny K6 propose/accept kernels.  This file is compiled with
// -ffp-contract=off: K1 must round after every operation to match the reference bit for bit.
#include "philox.h"
#include "si_internal.h"

namespace si {

// ------------------------------------------------------------------------------------------------
// K1  reference src/subspace_construction.jl:46-47,51-52
//       W_swa = (n.*W_swa + W)./(n+1)      -> t = n*s (rounded); u = t + w (rounded); s' = u/(n+1)
//       W_dev = W - W_swa ; append!(A, W_dev)
// Algorithmic bytes per element: sizeof(w) + 8 (read s) + 8 (write s) + 8 (write A column).
// Layout: s and the A column are 256-B aligned (padded leading dimension), so 16-B accesses are legal;
// each thread handles 4 consecutive elements per iteration (16 B of f32 w / 2 x 16 B of f64).
// ------------------------------------------------------------------------------------------------
// AT = element type of the deviation matrix: double (the reference: A = Array{Float64}, src/subspace_construction.jl:33,51-52) or
// float (the opt-in storage of SURVEY section 0 Q6: the column w - W_swa is formed in fp64 and rounded ONCE; W_swa itself stays fp64)
template <typename WT, typename AT>
__device__ __forceinline__ void push_one(const WT* __restrict__ w, double* __restrict__ s,
                                         AT* __restrict__ acol, int64_t i, double n, double np1) {
#pragma clang fp contract(off)
  const double wv = (double)w[i];
  const double t = n * s[i];
  const double u = t + wv;
  const double sn = u / np1;
  s[i] = sn;
  acol[i] = (AT)(wv - sn);
}

template <typename WT, bool W_ALIGNED, typename AT>
__global__ __launch_bounds__(256) void swa_dev_push_kernel(const WT* __restrict__ w,
                                                           double* __restrict__ s,
                                                           AT* __restrict__ acol, int64_t N,
                                                           double n, double np1) {
#pragma clang fp contract(off)
  const int64_t nquad = N >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += stride) {
    const int64_t i = q << 2;
    double wv[4];
    if constexpr (W_ALIGNED) {
      if constexpr (sizeof(WT) == 4) {
        const float4 f = *reinterpret_cast<const float4*>(w + i);
        wv[0] = (double)f.x; wv[1] = (double)f.y; wv[2] = (double)f.z; wv[3] = (double)f.w;
      } else {
        const double2 a = *reinterpret_cast<const double2*>(w + i);
        const double2 b = *reinterpret_cast<const double2*>(w + i + 2);
        wv[0] = a.x; wv[1] = a.y; wv[2] = b.x; wv[3] = b.y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) wv[j] = (double)w[i + j];
    }
    const double2 s0 = *reinterpret_cast<const double2*>(s + i);
    const double2 s1 = *reinterpret_cast<const double2*>(s + i + 2);
    const double sv[4] = {s0.x, s0.y, s1.x, s1.y};
    double sn[4], dv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double t = n * sv[j];
      const double u = t + wv[j];
      sn[j] = u / np1;
      dv[j] = wv[j] - sn[j];
    }
    *reinterpret_cast<double2*>(s + i) = make_double2(sn[0], sn[1]);
    *reinterpret_cast<double2*>(s + i + 2) = make_double2(sn[2], sn[3]);
    if constexpr (sizeof(AT) == 8) {
      *reinterpret_cast<double2*>(acol + i) = make_double2(dv[0], dv[1]);
      *reinterpret_cast<double2*>(acol + i + 2) = make_double2(dv[2], dv[3]);
    } else {
      *reinterpret_cast<float4*>(acol + i) = make_float4((float)dv[0], (float)dv[1], (float)dv[2], (float)dv[3]);
    }
  }
  // tail (N mod 4 elements) by the first threads of block 0
  if (blockIdx.x == 0) {
    const int64_t i = (nquad << 2) + threadIdx.x;
    if (i < N) push_one<WT, AT>(w, s, acol, i, n, np1);
  }
}

// K1, batched: `count` snapshots that are already device-resident (w_j = w + j*ld) are pushed in ONE pass.  W_swa
// stays in registers across the pushes, so the traffic per element drops from count*(s_w+24) to count*(s_w+8)+16 bytes
// and `count` launches become one.  Same three rounded operations per push, in the same order: bit-identical to
// `count` calls of swa_dev_push_kernel.  slot_j = (slot0 + j) mod kcap (ring only when max_cols is used).
template <typename WT, bool VEC, typename AT>
__global__ __launch_bounds__(256) void swa_dev_push_batch_kernel(const WT* __restrict__ w, int64_t ld,
                                                                 double* __restrict__ s, AT* __restrict__ A,
                                                                 int64_t ldA, int64_t N, int count,
                                                                 const double* __restrict__ nvals, int64_t slot0,
                                                                 int64_t kcap) {
#pragma clang fp contract(off)
  const int64_t npair = (N + 1) >> 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += stride) {
    const int64_t i = p << 1;
    const bool two = i + 1 < N;
    double2 sv = *reinterpret_cast<const double2*>(s + i);  // s is padded: reading s[N] is legal
    for (int j = 0; j < count; ++j) {
      const WT* wj = w + (int64_t)j * ld + i;
      double w0, w1;
      if constexpr (VEC) {
        if constexpr (sizeof(WT) == 4) {
          const float2 f = *reinterpret_cast<const float2*>(wj);
          w0 = (double)f.x; w1 = (double)f.y;
        } else {
          const double2 f = *reinterpret_cast<const double2*>(wj);
          w0 = f.x; w1 = f.y;
        }
      } else {
        w0 = (double)wj[0];
        w1 = two ? (double)wj[1] : 0.0;
      }
      const double n = nvals[j], np1 = n + 1.0;
      const double t0 = n * sv.x, t1 = n * sv.y;
      const double u0 = t0 + w0, u1 = t1 + w1;
      sv.x = u0 / np1;
      sv.y = u1 / np1;
      AT* col = A + ((slot0 + j) % kcap) * ldA + i;
      if (two) {
        if constexpr (sizeof(AT) == 8)
          *reinterpret_cast<double2*>(col) = make_double2(w0 - sv.x, w1 - sv.y);
        else
          *reinterpret_cast<float2*>(col) = make_float2((float)(w0 - sv.x), (float)(w1 - sv.y));
      } else {
        col[0] = (AT)(w0 - sv.x);
      }
    }
    if (two)
      *reinterpret_cast<double2*>(s + i) = sv;
    else
      s[i] = sv.x;
  }
}

static int stream_grid(int64_t work_items, int num_cu) {
  int64_t blocks = (work_items + 255) / 256;
  const int64_t cap = (int64_t)num_cu * 8;  // ~2048 blocks, grid-stride beyond (guide: Guideline 11)
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <typename AT>
static void launch_swa_dev_push_t(hipStream_t st, const void* w, int32_t w_dtype, double* s, AT* acol, int64_t N, double n, int num_cu) {
  const int grid = stream_grid(N >> 2, num_cu);
  const bool aligned = (reinterpret_cast<uintptr_t>(w) & 15u) == 0;
  const double np1 = n + 1.0;
  if (w_dtype == SI_F32) {
    const float* wf = static_cast<const float*>(w);
    if (aligned)
      hipLaunchKernelGGL((swa_dev_push_kernel<float, true, AT>), dim3(grid), dim3(256), 0, st, wf, s, acol, N, n, np1);
    else
      hipLaunchKernelGGL((swa_dev_push_kernel<float, false, AT>), dim3(grid), dim3(256), 0, st, wf, s, acol, N, n, np1);
  } else {
    const double* wd = static_cast<const double*>(w);
    if (aligned)
      hipLaunchKernelGGL((swa_dev_push_kernel<double, true, AT>), dim3(grid), dim3(256), 0, st, wd, s, acol, N, n, np1);
    else
      hipLaunchKernelGGL((swa_dev_push_kernel<double, false, AT>), dim3(grid), dim3(256), 0, st, wd, s, acol, N, n, np1);
  }
}
void launch_swa_dev_push(hipStream_t st, const void* w, int32_t w_dtype, double* s, void* acol, int64_t N, double n, int num_cu,
                         int32_t a_dtype) {
  if (a_dtype == SI_F32)
    launch_swa_dev_push_t<float>(st, w, w_dtype, s, static_cast<float*>(acol), N, n, num_cu);
  else
    launch_swa_dev_push_t<double>(st, w, w_dtype, s, static_cast<double*>(acol), N, n, num_cu);
}

template <typename AT>
static void launch_swa_dev_push_batch_t(hipStream_t st, const void* w, int32_t w_dtype, int64_t ld, double* s, AT* A, int64_t ldA,
                                        int64_t N, int count, const double* nvals_dev, int64_t slot0, int64_t kcap, int num_cu) {
  const int grid = stream_grid((N + 1) >> 1, num_cu);
  const size_t esz = w_dtype == SI_F32 ? 4 : 8;
  // vector loads need every snapshot row 2-element aligned and the last pair of an odd N inside the padded row
  const bool vec = (ld % 2 == 0) && (ld >= N + (N & 1)) && ((reinterpret_cast<uintptr_t>(w) & (2 * esz - 1)) == 0);
  if (w_dtype == SI_F32) {
    const float* wf = static_cast<const float*>(w);
    if (vec)
      hipLaunchKernelGGL((swa_dev_push_batch_kernel<float, true, AT>), dim3(grid), dim3(256), 0, st, wf, ld, s, A, ldA, N, count, nvals_dev, slot0, kcap);
    else
      hipLaunchKernelGGL((swa_dev_push_batch_kernel<float, false, AT>), dim3(grid), dim3(256), 0, st, wf, ld, s, A, ldA, N, count, nvals_dev, slot0, kcap);
  } else {
    const double* wd = static_cast<const double*>(w);
    if (vec)
      hipLaunchKernelGGL((swa_dev_push_batch_kernel<double, true, AT>), dim3(grid), dim3(256), 0, st, wd, ld, s, A, ldA, N, count, nvals_dev, slot0, kcap);
    else
      hipLaunchKernelGGL((swa_dev_push_batch_kernel<double, false, AT>), dim3(grid), dim3(256), 0, st, wd, ld, s, A, ldA, N, count, nvals_dev, slot0, kcap);
  }
}
void launch_swa_dev_push_batch(hipStream_t st, const void* w, int32_t w_dtype, int64_t ld, double* s, void* A, int64_t ldA, int64_t N,
                               int count, const double* nvals_dev, int64_t slot0, int64_t kcap, int num_cu, int32_t a_dtype) {
  if (a_dtype == SI_F32)
    launch_swa_dev_push_batch_t<float>(st, w, w_dtype, ld, s, static_cast<float*>(A), ldA, N, count, nvals_dev, slot0, kcap, num_cu);
  else
    launch_swa_dev_push_batch_t<double>(st, w, w_dtype, ld, s, static_cast<double*>(A), ldA, N, count, nvals_dev, slot0, kcap, num_cu);
}

// ------------------------------------------------------------------------------------------------
// K4  reference src/space_inference.jl:91 and :125   new_W = W_swa + P*z
// Algorithmic bytes per row: 8*(M+1) read + 8*C written; P is read ONCE for up to CB chains.
// Each thread owns 2 consecutive rows (16-B loads down each column of P: lanes are consecutive rows, so
// a wave reads 1 KiB contiguous per column); z is wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------------------
template <int CB>
__global__ __launch_bounds__(256) void reconstruct_kernel(const double* __restrict__ swa,
                                                          const double* __restrict__ P, int64_t ldP,
                                                          int64_t N, int32_t M,
                                                          const double* __restrict__ Z,
                                                          double* __restrict__ w, int64_t ldw,
                                                          float* __restrict__ w32, int64_t ldw32) {
  // w32 != nullptr (compute_dtype = SI_F32): the same fp64 sum is ALSO stored rounded once to fp32 -- the weights the
  // fp32 forward multiplies with; the fp64 copy keeps feeding the output map, the prior term and the gradient path
  const int64_t npair = (N + 1) >> 1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (blockIdx.y != 0) {   // chain groups of CB stacked in grid.y: one launch for all of them
    const int64_t cg = (int64_t)blockIdx.y * CB;
    Z += cg * M;
    w += cg * ldw;
    if (w32 != nullptr) w32 += cg * ldw32;
  }
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += stride) {
    const int64_t r = p << 1;
    double2 acc[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[c] = make_double2(0.0, 0.0);
    for (int m = 0; m < M; ++m) {
      const double2 pv = *reinterpret_cast<const double2*>(P + r + (int64_t)m * ldP);
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const double z = Z[m + c * M];
        acc[c].x += pv.x * z;
        acc[c].y += pv.y * z;
      }
    }
    const double2 sv = *reinterpret_cast<const double2*>(swa + r);
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      double* dst = w + (int64_t)c * ldw + r;
      if (r + 1 < N) {
        if ((ldw & 1) == 0)
          *reinterpret_cast<double2*>(dst) = make_double2(sv.x + acc[c].x, sv.y + acc[c].y);
        else {
          dst[0] = sv.x + acc[c].x;
          dst[1] = sv.y + acc[c].y;
        }
      } else {
        dst[0] = sv.x + acc[c].x;
      }
      if (w32 != nullptr) {   // ldw32 is even (pad_ld) and the rows come in pairs: 8-byte stores
        float* d32 = w32 + (int64_t)c * ldw32 + r;
        if (r + 1 < N)
          *reinterpret_cast<float2*>(d32) = make_float2((float)(sv.x + acc[c].x), (float)(sv.y + acc[c].y));
        else
          d32[0] = (float)(sv.x + acc[c].x);
      }
    }
  }
}

void launch_reconstruct(hipStream_t st, const double* swa, const double* P, int64_t ldP, int64_t N,
                        int32_t M, const double* Z, int32_t C, double* w, int64_t ldw, int num_cu, float* w32, int64_t ldw32) {
  const int grid = stream_grid((N + 1) >> 1, num_cu);
  int c0 = 0;
  if (C >= 8) {   // many stacked chains: every group of four in one launch (512 chains were 128 dependent launches)
    const int n4 = C / 4;
    hipLaunchKernelGGL((reconstruct_kernel<4>), dim3(grid, n4), dim3(256), 0, st, swa, P, ldP, N, M, Z, w, ldw, w32, ldw32);
    c0 = 4 * n4;
  }
  while (c0 < C) {
    const int rem = C - c0;
    const double* Zc = Z + (int64_t)c0 * M;
    double* wc = w + (int64_t)c0 * ldw;
    float* wc32 = w32 ? w32 + (int64_t)c0 * ldw32 : nullptr;
    if (rem >= 4) {
      hipLaunchKernelGGL((reconstruct_kernel<4>), dim3(grid), dim3(256), 0, st, swa, P, ldP, N, M, Zc, wc, ldw, wc32, ldw32);
      c0 += 4;
    } else if (rem >= 2) {
      hipLaunchKernelGGL((reconstruct_kernel<2>), dim3(grid), dim3(256), 0, st, swa, P, ldP, N, M, Zc, wc, ldw, wc32, ldw32);
      c0 += 2;
    } else {
      hipLaunchKernelGGL((reconstruct_kernel<1>), dim3(grid), dim3(256), 0, st, swa, P, ldP, N, M, Zc, wc, ldw, wc32, ldw32);
      c0 += 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K5 tail  reference src/space_inference.jl:94   ||vec(Y) - vec(f(X))||^2
// Deterministic: fixed block count, wave shuffle -> LDS -> per-block partial; block 0 of a second launch
// sums the partials in index order.  Same inputs => same bits.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void sse_partial_kernel(const double* __restrict__ yhat,
                                                          const double* __restrict__ y, int64_t d,
                                                          double* __restrict__ part, int64_t yhat_stride) {
  __shared__ double red[4];
  double acc = 0.0;
  yhat += (int64_t)blockIdx.y * yhat_stride;  // chain slot
  part += (int64_t)blockIdx.y * gridDim.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += stride) {
    const double r = (y ? y[i] : 0.0) - yhat[i];   // y == nullptr: plain sum of squares (the optional prior term ||w||^2)
    acc += r * r;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sse_final_kernel(const double* __restrict__ part, int n,
                                                        double* __restrict__ out) {
  __shared__ double red[4];
  double acc = 0.0;
  part += (int64_t)blockIdx.x * n;  // one block per chain slot
  for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

int sse_num_blocks(int64_t d, int num_cu) {
  int64_t b = (d + 255) / 256;   // one element per thread up to 4 blocks per CU (the tail of a step is latency-bound)
  if (b > (int64_t)num_cu * 4) b = (int64_t)num_cu * 4;
  if (b < 1) b = 1;
  return (int)b;
}

void launch_sse_final(hipStream_t st, const double* blockpart, int nblocks, double* sse_out, int nch) {
  hipLaunchKernelGGL(sse_final_kernel, dim3(nch), dim3(256), 0, st, blockpart, nblocks, sse_out);
}

void launch_sse(hipStream_t st, const double* yhat, const double* y, int64_t d, double* part,
                int nblocks, double* sse_out, int nch, int64_t yhat_stride, bool with_final) {
  hipLaunchKernelGGL(sse_partial_kernel, dim3(nblocks, nch), dim3(256), 0, st, yhat, y, d, part, yhat_stride);
  if (with_final) hipLaunchKernelGGL(sse_final_kernel, dim3(nch), dim3(256), 0, st, part, nblocks, sse_out);
}

// ------------------------------------------------------------------------------------------------
// K6  reference src/space_inference.jl:111-116 (AdvancedMH 0.6.2 RWMH + AbstractMCMC sample)
//   step 0: z0 ~ N(0, sigma_z^2 I), always kept;  step t: z' = z + sigma_z*eps, accept iff -Exp(1) < lp'-lp
// State stays on the device; one block per chain.
// ------------------------------------------------------------------------------------------------
__global__ void rwmh_init_kernel(double* zcur, double* lpcur, int64_t* nacc, uint64_t* steps, int32_t M) {
  const int c = blockIdx.x;
  for (int m = threadIdx.x; m < M; m += blockDim.x) zcur[m + c * M] = 0.0;
  if (threadIdx.x == 0) {
    lpcur[c] = -__builtin_inf();
    nacc[c] = 0;
    steps[c] = 0;
  }
}

__global__ void rwmh_propose_kernel(const double* __restrict__ zcur, double* __restrict__ zprop,
                                    int32_t M, double sigma_z, uint64_t seed, int32_t chain_id0,
                                    const uint64_t* __restrict__ steps) {
  // the transition index lives on the device (one counter per chain, advanced by the accept kernel): the kernel
  // arguments of a transition never change, so a captured hipGraph of one transition can be replayed itr-1 times
  const int c = blockIdx.x;
  const uint64_t step = steps[c];
  const uint32_t chain = (uint32_t)(chain_id0 + c);
  const int nblk = (M + 1) >> 1;
  for (int j = threadIdx.x; j < nblk; j += blockDim.x) {
    double n0, n1;
    philox_normal2(seed, chain, step, (uint32_t)j, n0, n1);
    const int m0 = 2 * j;
    zprop[m0 + c * M] = zcur[m0 + c * M] + sigma_z * n0;
    if (m0 + 1 < M) zprop[m0 + 1 + c * M] = zcur[m0 + 1 + c * M] + sigma_z * n1;
  }
}

__global__ void rwmh_accept_kernel(double* __restrict__ zcur, const double* __restrict__ zprop,
                                   double* __restrict__ lpcur, const double* __restrict__ sse,
                                   int64_t* __restrict__ nacc, int32_t M, double c0, double sigma2,
                                   uint64_t seed, int32_t chain_id0, uint64_t* __restrict__ steps,
                                   double* __restrict__ Z_out, double* __restrict__ lp_out,
                                   int64_t itr, const double* __restrict__ wsq, double c0p, double sigma_p2,
                                   int32_t* __restrict__ accflag) {
  const int c = blockIdx.x;
  const uint64_t step = steps[c];
  const uint32_t chain = (uint32_t)(chain_id0 + c);
  // Distributions.logpdf(MvNormal(mu, sigma), y) = c0 - (sse / sigma^2) / 2
  double lp_new = c0 - (sse[c] / sigma2) / 2.0;
  // non-default option: + logpdf(MvNormal(zeros(N), sigma_p), new_W) -- the term the reference leaves dead (Q4)
  if (wsq) lp_new += c0p - (wsq[c] / sigma_p2) / 2.0;
  const double lp_old = lpcur[c];
  bool accept;
  if (step == 0) {
    accept = true;
  } else {
    const double e = philox_randexp(seed, chain, step);
    accept = (-e < lp_new - lp_old);  // NaN compares false => reject, as in Julia
  }
  const double lp_keep = accept ? lp_new : lp_old;
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    const double zv = accept ? zprop[m + c * M] : zcur[m + c * M];
    zcur[m + c * M] = zv;
    Z_out[m + (int64_t)M * (step + (uint64_t)itr * c)] = zv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lpcur[c] = lp_keep;
    lp_out[step + (uint64_t)itr * c] = lp_keep;
    if (accept && step > 0) nacc[c] += 1;
    steps[c] = step + 1;
    if (accflag) accflag[c] = accept ? 1 : 0;
  }
}

// The tail of a transition in ONE launch (the fused sampler loop, all chains in one pass of launches): the last stage of the
// SSE reduction (sse_final_kernel: same 256-thread sweep, same wave sums, same (r0 + r1) + (r2 + r3)), the accept step
// (rwmh_accept_kernel) and the NEXT transition's proposal (rwmh_propose_kernel) -- three dependent 5-us launches less per
// transition, which is a fifth of a transition for a model of the size of docs/src/nn_example.md.  Same functions in the same
// order: the chain is bit-identical to the three-kernel form (tests/test_gpu_chain.py keeps both).
__global__ __launch_bounds__(256) void rwmh_tail_kernel(const double* __restrict__ ssepart, int nparts, double* __restrict__ sse,
                                                        double* __restrict__ zcur, double* __restrict__ zprop,
                                                        double* __restrict__ lpcur, int64_t* __restrict__ nacc, int32_t M, double c0,
                                                        double sigma2, double sigma_z, uint64_t seed, int32_t chain_id0,
                                                        uint64_t* __restrict__ steps, double* __restrict__ Z_out,
                                                        double* __restrict__ lp_out, int64_t itr, int32_t* __restrict__ accflag,
                                                        int propose_next) {
  __shared__ double red[4];
  __shared__ double sse_s;
  const int c = blockIdx.x;
  {
    double acc = 0.0;
    const double* part = ssepart + (int64_t)c * nparts;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double t = (red[0] + red[1]) + (red[2] + red[3]);
      sse[c] = t;
      sse_s = t;
    }
    __syncthreads();
  }
  const uint64_t step = steps[c];
  const uint32_t chain = (uint32_t)(chain_id0 + c);
  const double lp_new = c0 - (sse_s / sigma2) / 2.0;
  const double lp_old = lpcur[c];
  bool accept;
  if (step == 0) {
    accept = true;
  } else {
    const double e = philox_randexp(seed, chain, step);
    accept = (-e < lp_new - lp_old);  // NaN compares false => reject, as in Julia
  }
  const double lp_keep = accept ? lp_new : lp_old;
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    const double zv = accept ? zprop[m + c * M] : zcur[m + c * M];
    zcur[m + c * M] = zv;
    Z_out[m + (int64_t)M * (step + (uint64_t)itr * c)] = zv;
  }
  __syncthreads();   // zcur complete (and every thread has read steps / lpcur) before thread 0 advances them
  if (threadIdx.x == 0) {
    lpcur[c] = lp_keep;
    lp_out[step + (uint64_t)itr * c] = lp_keep;
    if (accept && step > 0) nacc[c] += 1;
    steps[c] = step + 1;
    if (accflag) accflag[c] = accept ? 1 : 0;
  }
  if (propose_next) {   // rwmh_propose_kernel of transition step + 1 on the state just written
    const int nblk = (M + 1) >> 1;
    for (int j = threadIdx.x; j < nblk; j += blockDim.x) {
      double n0, n1;
      philox_normal2(seed, chain, step + 1, (uint32_t)j, n0, n1);
      const int m0 = 2 * j;
      zprop[m0 + c * M] = zcur[m0 + c * M] + sigma_z * n0;
      if (m0 + 1 < M) zprop[m0 + 1 + c * M] = zcur[m0 + 1 + c * M] + sigma_z * n1;
    }
  }
}

// Output map (src/space_inference.jl:125) without a second K4 pass: K4 already produced W_swa + P z' for the proposal;
// the chain's current weights are that vector when the proposal was kept, the previous sample's otherwise.
__global__ __launch_bounds__(256) void weights_select_kernel(const int32_t* __restrict__ flag, const double* __restrict__ wprop,
                                                             int64_t ldw, const double* __restrict__ prev, double* __restrict__ dst,
                                                             int64_t ldd, int64_t N) {
  const int c = blockIdx.y;
  const double* src = (prev == nullptr || flag[c] != 0) ? wprop + (int64_t)c * ldw : prev + (int64_t)c * ldd;
  double* out = dst + (int64_t)c * ldd;
  const int64_t npair = (N + 1) >> 1;   // both buffers are padded to an even length (pad_ld)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += stride)
    *reinterpret_cast<double2*>(out + 2 * p) = *reinterpret_cast<const double2*>(src + 2 * p);
}

void launch_rwmh_init(hipStream_t st, double* zcur, double* lpcur, int64_t* nacc, uint64_t* steps, int32_t M, int32_t C) {
  hipLaunchKernelGGL(rwmh_init_kernel, dim3(C), dim3(64), 0, st, zcur, lpcur, nacc, steps, M);
}
void launch_rwmh_propose(hipStream_t st, const double* zcur, double* zprop, int32_t M, int32_t C,
                         double sigma_z, uint64_t seed, int32_t chain_id0, const uint64_t* steps) {
  hipLaunchKernelGGL(rwmh_propose_kernel, dim3(C), dim3(64), 0, st, zcur, zprop, M, sigma_z, seed, chain_id0, steps);
}
void launch_rwmh_accept(hipStream_t st, double* zcur, const double* zprop, double* lpcur,
                        const double* sse, int64_t* nacc, int32_t M, int32_t C, double c0,
                        double sigma2, uint64_t seed, int32_t chain_id0, uint64_t* steps,
                        double* Z_out, double* lp_out, int64_t itr, const double* wsq, double c0p, double sigma_p2,
                        int32_t* accflag) {
  hipLaunchKernelGGL(rwmh_accept_kernel, dim3(C), dim3(64), 0, st, zcur, zprop, lpcur, sse, nacc, M, c0,
                     sigma2, seed, chain_id0, steps, Z_out, lp_out, itr, wsq, c0p, sigma_p2, accflag);
}
void launch_rwmh_tail(hipStream_t st, const double* ssepart, int nparts, double* sse, double* zcur, double* zprop, double* lpcur,
                      int64_t* nacc, int32_t M, int32_t C, double c0, double sigma2, double sigma_z, uint64_t seed, int32_t chain_id0,
                      uint64_t* steps, double* Z_out, double* lp_out, int64_t itr, int32_t* accflag, bool propose_next) {
  hipLaunchKernelGGL(rwmh_tail_kernel, dim3(C), dim3(256), 0, st, ssepart, nparts, sse, zcur, zprop, lpcur, nacc, M, c0, sigma2, sigma_z,
                     seed, chain_id0, steps, Z_out, lp_out, itr, accflag, propose_next ? 1 : 0);
}
void launch_weights_select(hipStream_t st, const int32_t* flag, const double* wprop, int64_t ldw, const double* prev, double* dst,
                           int64_t ldd, int64_t N, int32_t C, int num_cu) {
  hipLaunchKernelGGL(weights_select_kernel, dim3(stream_grid((N + 1) >> 1, num_cu), C), dim3(256), 0, st, flag, wprop, ldw, prev, dst,
                     ldd, N);
}

// g[i] -= w[i] * inv_s2   (gradient of the optional prior term -||w||^2 / (2 sigma_p^2))
__global__ __launch_bounds__(256) void prior_grad_kernel(double* __restrict__ g, const double* __restrict__ w, int64_t n, double inv_s2) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] -= w[i] * inv_s2;
}
void launch_prior_grad(hipStream_t st, double* g, const double* w, int64_t n, double inv_s2, int num_cu) {
  hipLaunchKernelGGL(prior_grad_kernel, dim3(stream_grid(n, num_cu)), dim3(256), 0, st, g, w, n, inv_s2);
}
// At[k + ldt*r] = A[r + lda*k]: the deviation matrix transposed, for the K > N route of si_construct_finish (the Gram kernel
// reads columns: the Gram matrix of A' is A A').  32 x 32 tiles through LDS, both sides coalesced.
template <typename AT>
__global__ __launch_bounds__(256) void transpose_kernel(const AT* __restrict__ A, int64_t lda, int64_t N, int64_t K,
                                                        double* __restrict__ At, int64_t ldt) {
  __shared__ double tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int64_t r = r0 + tx, k = k0 + j;
    tile[j][tx] = (r < N && k < K) ? (double)A[r + lda * k] : 0.0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int64_t k = k0 + tx, r = r0 + j;
    if (k < K && r < N) At[k + ldt * r] = tile[tx][j];
  }
}
void launch_transpose(hipStream_t st, const void* A, int32_t a_dtype, int64_t lda, int64_t N, int64_t K, double* At, int64_t ldt) {
  const dim3 grid((unsigned)((N + 31) / 32), (unsigned)((K + 31) / 32));
  if (a_dtype == SI_F32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, st, static_cast<const float*>(A), lda, N, K, At, ldt);
  else
    hipLaunchKernelGGL(transpose_kernel<double>, grid, dim3(256), 0, st, static_cast<const double*>(A), lda, N, K, At, ldt);
}

// dst[i] = (double)src[i]  (initial W_swa from Float32 weights: the non-default init = :pretrained option)
__global__ __launch_bounds__(256) void widen_f32_kernel(const float* __restrict__ src, double* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (double)src[i];
}
void launch_widen_f32(hipStream_t st, const float* src, double* dst, int64_t n, int num_cu) {
  hipLaunchKernelGGL(widen_f32_kernel, dim3(stream_grid(n, num_cu)), dim3(256), 0, st, src, dst, n);
}

}  // namespace si
