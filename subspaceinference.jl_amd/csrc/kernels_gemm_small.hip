// K5 for SMALL layers (round 4): the same operation as kernels_gemm.hip -- reference src/space_inference.jl:92-94, per Flux
// Dense layer H' = act.(W*H .+ b) -- for shapes on which the big-tile kernel is all latency.  docs/src/nn_example.md's MLP
// (2-200-50-50-50-1, B = 1000) gives dense_f64_kernel's 64 x 128 tiles 8 workgroups per layer, each exposing a full
// global-load latency per 16-deep k tile: 10-14 us per layer, 49 of the 60 us of a transition (rocprofv3 kernel trace, DESIGN
// 9.10).  Here ONE WAVE owns one (feature slot) x (16 observations) tile and takes its MFMA operands straight from global
// memory -- lane (q, c) loads W[i0 + 16a + c][4s + q] and H[4s + q][b0 + c], the layout v_mfma_f64_16x16x4_f64 wants -- in
// chunks of k steps, two chunks in flight (double-buffered registers: one wave per SIMD may use the whole register file), no
// LDS, no barrier.  A 50 x 1000 layer is 126 independent waves.
//
// SAME BITS as dense_f64_kernel and as the device-resident loop of kernels_chain.hip: every output element is accumulated
// over k in the same order by the same instruction (k steps of 4 in ascending order, the k range padded with zeros to whole
// 16-deep tiles), bias and activation are applied the same way, and the fused head uses the same feature slots
// (dense_fused_slot_feats: 32 / 48 / 64 rows = TM 16-row tiles), the same fma chain over a slot's tiles and the same
// 16-lane butterfly.  Which kernel a layer gets is a function of its shape alone (dense_small_applies).
#include "kernels_gemm.h"

namespace si {

// (body and kernel apart: a __global__ template with device builtins inside lambdas loses its host stub)
template <int TM, bool FUSE>
__device__ __forceinline__ void dense_small_f64_body(const double* __restrict__ W, const double* __restrict__ bias,
                                                     const double* __restrict__ Hin, double* __restrict__ Hout, int out, int in,
                                                     int64_t B, int act, int nslots, const double* __restrict__ Wlast,
                                                     int out_last, double* __restrict__ part, const ChainBatch& cb) {
  if (blockIdx.y != 0) {   // chain batching: every operand that differs per chain moves by its slot stride
    const int64_t ch = blockIdx.y;
    W += ch * cb.w;
    bias += ch * cb.w;
    Hin += ch * cb.hin;
    if constexpr (FUSE) {
      Wlast += ch * cb.w;
      part += ch * cb.part;
      if (Hout != nullptr) Hout += ch * cb.hout;
    } else {
      Hout += ch * cb.hout;
    }
  }
  constexpr int KC = 32 / TM;   // k steps (of 4) per register chunk: 16 / 10 / 8
  const int slot = (int)(blockIdx.x % (unsigned)nslots);
  const int64_t jb = blockIdx.x / (unsigned)nslots;
  const int lane = threadIdx.x, q = lane >> 4, c = lane & 15;
  const int i0 = slot * (TM * 16);
  const int64_t b0 = jb * 16;

  int gi[TM];
  const double* wp[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    gi[a] = i0 + 16 * a + c;
    wp[a] = W + (gi[a] < out ? gi[a] : out - 1);   // rows past `out` only feed outputs that are never stored
  }
  const int64_t gcol = b0 + c < B ? b0 + c : B - 1;
  const double* hp = Hin + (int64_t)in * gcol;
  const int nsteps = (in + 15) / 16 * 4;           // the big kernel's k range: whole 16-deep tiles, zeros past `in`
  const int nchunk = (nsteps + KC - 1) / KC;

  d4 acc[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) acc[a] = (d4){0.0, 0.0, 0.0, 0.0};

  double fa0[TM][KC], fb0[KC], fa1[TM][KC], fb1[KC];
  auto load = [&](int ch, double (&fa)[TM][KC], double (&fb)[KC]) {
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      const int k = 4 * (ch * KC + s) + q;
      const bool ok = k < in;
      const int kk = ok ? k : 0;               // the address is always a valid one; the value past `in` is zero
      const double hv = hp[kk];
      fb[s] = ok ? hv : 0.0;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const double wv = wp[a][(int64_t)out * kk];
        fa[a][s] = ok ? wv : 0.0;
      }
    }
  };
  auto compute = [&](int ch, const double (&fa)[TM][KC], const double (&fb)[KC]) {
#pragma unroll
    for (int s = 0; s < KC; ++s) {
      if (ch * KC + s < nsteps) {   // wave-uniform
#pragma unroll
        for (int a = 0; a < TM; ++a) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[s], fa[a][s], acc[a], 0, 0, 0);
      }
    }
  };
  load(0, fa0, fb0);
  for (int ch = 0; ch < nchunk; ch += 2) {
    if (ch + 1 < nchunk) load(ch + 1, fa1, fb1);
    compute(ch, fa0, fb0);
    if (ch + 2 < nchunk) load(ch + 2, fa0, fb0);
    if (ch + 1 < nchunk) compute(ch + 1, fa1, fb1);
  }

  // ---- epilogue (kernels_gemm.hip, TN = 1): a lane holds D[b = q + 4r][i = c] of each of its TM tiles
  auto finish = [&](double v) -> double {
    if (act == SI_ACT_RELU) return v > 0.0 ? v : 0.0;
    if (act == SI_ACT_IDENTITY) return v;
    return act_full(v, act);
  };
  double bv[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) bv[a] = gi[a] < out ? bias[gi[a]] : 0.0;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[a][r] = finish(acc[a][r] + bv[a]);
  if (!FUSE || Hout != nullptr) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gb = b0 + q + 4 * r;
        if (gi[a] < out && gb < B) Hout[gi[a] + (int64_t)out * gb] = acc[a][r];
      }
  }
  if constexpr (FUSE) {
    for (int o = 0; o < out_last; ++o) {
      double wl[TM];
#pragma unroll
      for (int a = 0; a < TM; ++a) wl[a] = gi[a] < out ? Wlast[o + (int64_t)out_last * gi[a]] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double p = 0.0;
#pragma unroll
        for (int a = 0; a < TM; ++a) p = fma(acc[a][r], wl[a], p);
        // the 16 lanes of a 16-lane row hold the 16 features of one tile column: butterfly sum inside the row
        p += __shfl_xor(p, 8, 16);
        p += __shfl_xor(p, 4, 16);
        p += __shfl_xor(p, 2, 16);
        p += __shfl_xor(p, 1, 16);
        const int64_t gb = b0 + q + 4 * r;
        if (c == 0 && gb < B) part[((int64_t)slot * out_last + o) * (cb.part_ld ? cb.part_ld : B) + gb] = p;
      }
    }
  }
}

template <int TM, bool FUSE>
__global__ __launch_bounds__(64) void dense_small_f64_kernel(const double* __restrict__ W, const double* __restrict__ bias,
                                                            const double* __restrict__ Hin, double* __restrict__ Hout, int out,
                                                            int in, int64_t B, int act, int nslots,
                                                            const double* __restrict__ Wlast, int out_last,
                                                            double* __restrict__ part, ChainBatch cb) {
  dense_small_f64_body<TM, FUSE>(W, bias, Hin, Hout, out, in, B, act, nslots, Wlast, out_last, part, cb);
}

// The small kernel takes a layer when the big-tile kernel would start no more than a quarter of the chip's workgroup slots
// (its tiles are `bm` x 128) and the k range is short enough for the register chunks to matter.
bool dense_small_applies(int32_t out, int32_t in, int64_t B, int nchains, int32_t bm) {
  const int64_t big = (int64_t)((out + bm - 1) / bm) * ((B + 127) / 128) * (nchains > 0 ? nchains : 1);
  return big <= 128 && in <= 2048 && out >= 1 && B >= 1;
}

template <bool FUSE>
static void launch_small(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout, int32_t out, int32_t in,
                         int64_t B, int32_t act, int32_t slot_feats, int32_t nslots, const double* Wlast, int32_t out_last, double* part,
                         const ChainBatch& cb) {
  const int64_t ncol = (B + 15) / 16;
  const dim3 grid((unsigned)(nslots * ncol), (unsigned)cb.n);
  switch (slot_feats / 16) {
    case 2: hipLaunchKernelGGL((dense_small_f64_kernel<2, FUSE>), grid, dim3(64), 0, st, W, bias, Hin, Hout, (int)out, (int)in, B, (int)act, nslots, Wlast, (int)out_last, part, cb); break;
    case 3: hipLaunchKernelGGL((dense_small_f64_kernel<3, FUSE>), grid, dim3(64), 0, st, W, bias, Hin, Hout, (int)out, (int)in, B, (int)act, nslots, Wlast, (int)out_last, part, cb); break;
    default: hipLaunchKernelGGL((dense_small_f64_kernel<4, FUSE>), grid, dim3(64), 0, st, W, bias, Hin, Hout, (int)out, (int)in, B, (int)act, nslots, Wlast, (int)out_last, part, cb); break;
  }
}

void launch_dense_small_f64(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout, int32_t out,
                            int32_t in, int64_t B, int32_t act, int32_t slot_feats, const ChainBatch& cb) {
  launch_small<false>(st, W, bias, Hin, Hout, out, in, B, act, slot_feats, (out + slot_feats - 1) / slot_feats, nullptr, 0, nullptr, cb);
}
void launch_dense_small_f64_fused(hipStream_t st, const double* W, const double* bias, const double* Hin, int32_t out, int32_t in,
                                  int64_t B, int32_t act, int32_t slot_feats, int32_t nslots, const double* Wlast, int32_t out_last,
                                  double* part, const ChainBatch& cb, double* Hkeep) {
  // nslots = dense_fused_slots(out), the row count of `part` the tail sums: slots past the last feature write zeros, as the
  // padded rows of a big tile do
  launch_small<true>(st, W, bias, Hin, Hkeep, out, in, B, act, slot_feats, nslots, Wlast, out_last, part, cb);
}

}  // namespace si
