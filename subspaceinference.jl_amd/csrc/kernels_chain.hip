// K6 as SURVEY section 2.2 specifies it: a DEVICE-RESIDENT RWMH loop for small Dense chains.  One workgroup per chain runs
// all `itr` transitions of reference src/space_inference.jl:111-116 (AdvancedMH RWMH: propose -> density -> accept) in ONE
// launch, with the reconstructed weights, the data, the activations and the head partials of its chain in LDS.
//
// Why: the launch-per-step path (capi_sample.hip sample_rwmh_impl) pays ~8 dependent kernel launches per transition; for the only
// workloads the reference documents (README.md:52-79: N = 682, M = 3, B = 100) that is 25 us of launch latency around
// ~1 us of arithmetic (profiles/r03_small_model_steps.log; a hipGraph replay did not help: the kernels are DEPENDENT).
//
// Bit-identical to the launch-per-step path by construction -- every number goes through the same operations in the same order:
//   propose   zprop = zcur + sigma_z * eps, Philox stream of csrc/philox.h                 (rwmh_propose_kernel)
//   K4        acc = 0; acc += P[r, m] * z[m] for m = 0..M-1 (mul, add); w = W_swa[r] + acc  (reconstruct_kernel)
//   layers    v_mfma_f64_16x16x4_f64 over k steps of 4 in increasing k (an all-zero k step adds nothing), A = H, B = W;
//             h = act(acc + bias)                                                         (dense_f64_kernel)
//   head      the narrow last layer inside the epilogue of the layer before it: per feature slot of `slot_feats` features
//             p = fma(h_a, wl_a, p) over the slot's 16-feature tiles in order, butterfly over the 16 lanes (xor 8, 4, 2, 1)
//   tail      sum of the slots in order + bias, activation, (y - yhat)^2; 256-thread virtual blocks: shuffle-down wave sums,
//             (r0 + r1) + (r2 + r3), then the block partials in index order                 (tail_sse_kernel, sse_final_kernel)
//   accept    lp' = c0 - (sse / sigma^2) / 2; accept iff step == 0 or -Exp(1) < lp' - lp    (rwmh_accept_kernel)
// This file is compiled with -ffp-contract=off like kernels_stream.hip (the only fused multiply-adds of the path are the
// explicit fma of the head and the MFMAs); tests/test_gpu_chain.py asserts array_equal against si_sample_rwmh's loop.
#include <algorithm>

#include "chain_common.h"
#include "philox.h"

namespace si {

#ifdef SI_CHAIN_STAMPS   // tools/chain_bench.hip only: cycles per phase of chain 0, summed over the transitions
#define SI_CSTAMP(i) do { if (tid == 0 && blockIdx.x == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); cst[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define SI_CSTAMP(i) do {} while (0)
#endif

typedef double cd4 __attribute__((ext_vector_type(4)));

// Everything the loop touches lives in LDS and is addressed as chain_lds[offset] (a `const double*` that may point to either of
// two LDS buffers degrades to a FLAT pointer).  Operands sit in ZERO-PADDED images so that a tile needs no bounds logic:
//   weights   per layer l < L-1:  Wp[i + outp*k], i < outp = ceil16(out), k < inp = ceil4(in); bias bp[i], i < outp
//             (K4 scatters W_swa + P z there through an index map built once; the padding is zeroed once and never written)
//   inputs    Hp[k + ldh*b], k < ldh = ceil4(in), b < Bp = ceil16(B); rows k >= in stay zero, columns b >= B hold whatever
//             (they only feed batch columns that are never stored)
extern __shared__ double chain_lds[];

// NS k steps of one 16 x 16 tile: all 2 NS operands are requested first (unconditional loads from the padded images, one
// LDS latency for the lot), then the NS MFMAs run as one dependent chain in increasing k
template <int NS>
__device__ __forceinline__ cd4 chain_steps(int wb, int hb, int wstep, cd4 acc) {
  double a[NS], b[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    a[j] = chain_lds[hb + 4 * j];
    b[j] = chain_lds[wb + wstep * j];
  }
#pragma unroll
  for (int j = 0; j < NS; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[j], acc, 0, 0, 0);
  return acc;
}

// one 16 x 16 output tile (features 16 mt .., batch 16 nt ..) of Wp * Hp: D[b = q + 4r][i = c]
__device__ __forceinline__ cd4 chain_tile(int Wo, int outp, int Ho, int ldh, int nsteps, int mt, int nt, int lane) {
  const int q = lane >> 4, c = lane & 15;
  int wb = Wo + 16 * mt + c + outp * q;         // + 4 * outp per k step
  int hb = Ho + q + ldh * (16 * nt + c);        // + 4 per k step
  const int wstep = 4 * outp;
  cd4 acc = {0.0, 0.0, 0.0, 0.0};
  int rem = nsteps;
  for (; rem >= 8; rem -= 8, wb += 8 * wstep, hb += 32) acc = chain_steps<8>(wb, hb, wstep, acc);
  switch (rem) {   // (uniform: one scalar branch)
    case 1: acc = chain_steps<1>(wb, hb, wstep, acc); break;
    case 2: acc = chain_steps<2>(wb, hb, wstep, acc); break;
    case 3: acc = chain_steps<3>(wb, hb, wstep, acc); break;
    case 4: acc = chain_steps<4>(wb, hb, wstep, acc); break;
    case 5: acc = chain_steps<5>(wb, hb, wstep, acc); break;
    case 6: acc = chain_steps<6>(wb, hb, wstep, acc); break;
    case 7: acc = chain_steps<7>(wb, hb, wstep, acc); break;
    default: break;
  }
  return acc;
}

// 16 waves per chain: a layer of the README toy is 14 tiles -- one round instead of four
constexpr int CT = 1024;

__global__ __launch_bounds__(CT) void rwmh_chain_kernel(ChainLoopArgs A) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int chain = blockIdx.x;
  const int N = A.N, M = A.M, B = A.B, L = A.L;
  const int Bp = (B + 15) & ~15;
  int* smap = reinterpret_cast<int*>(chain_lds + A.o_map);
  const si_layer& lf = A.lay[L - 2];                  // the layer whose epilogue carries the head
  const si_layer& ll = A.lay[L - 1];
  const int outL = ll.out, d = outL * B;
  // ---- once: zero the padded images, build K4's index map, stage W_swa, P (when it fits), X, Y
  for (int i = tid; i < A.o_map; i += CT) chain_lds[i] = 0.0;     // weights, X, activations, partials: everything in front of the map
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    const si_layer& ly = A.lay[l];
    const int outp = l < L - 1 ? (ly.out + 15) & ~15 : ly.out;
    const int nw = ly.in * ly.out;
    for (int e = tid; e < nw + ly.out; e += CT) {
      if (e < nw) smap[(int)ly.w_off + e] = A.wp[l] + (e % ly.out) + outp * (e / ly.out);
      else smap[(int)ly.b_off + e - nw] = A.bp[l] + e - nw;
    }
  }
  {
    const int in0 = A.lay[0].in, ld0 = (in0 + 3) & ~3;
    for (int i = tid; i < in0 * B; i += CT) chain_lds[A.o_X + (i % in0) + ld0 * (i / in0)] = A.X[i];
  }
  for (int i = tid; i < d; i += CT) chain_lds[A.o_Y + i] = A.Y[i];
  if (A.p_in_lds)
    for (int i = tid; i < N * M; i += CT) chain_lds[A.o_P + i] = A.P[(i % N) + (int64_t)A.ldP * (i / N)];
  for (int i = tid; i < N; i += CT) chain_lds[A.o_swa + i] = A.swa[i];
  if (tid < M) chain_lds[A.o_z + tid] = 0.0;            // zcur (rwmh_init_kernel)
  if (tid == 0) chain_lds[A.o_red + 5] = -__builtin_inf();
  int64_t nacc = 0;
  const int64_t zbase = (int64_t)M * A.itr * chain, lbase = A.itr * (int64_t)chain;
  const uint32_t chain_id = (uint32_t)(A.chain_id0 + chain);
  // ---- the random numbers of the whole chain, all threads at once: a draw depends on (seed, chain, step) only, and two
  // threads evaluating fp64 log / sqrt / sin / cos inside every transition would be a serial phase of it.  The proposal noise
  // of transition t waits in Z_out[:, t] (overwritten by the sample itself once t is decided), the Exp(1) of its acceptance
  // test in lp_out[t].  Same functions, same bits as rwmh_propose_kernel / rwmh_accept_kernel.
  const int nblk = (M + 1) >> 1;
  for (int64_t i = tid; i < A.itr * nblk; i += CT) {
    const int64_t t = i / nblk;
    const int j = (int)(i % nblk);
    double n0, n1;
    philox_normal2(A.seed, chain_id, (uint64_t)t, (uint32_t)j, n0, n1);
    A.Z_out[zbase + 2 * j + (int64_t)M * t] = n0;
    if (2 * j + 1 < M) A.Z_out[zbase + 2 * j + 1 + (int64_t)M * t] = n1;
  }
  for (int64_t t = tid; t < A.itr; t += CT) A.lp_out[lbase + t] = t == 0 ? 0.0 : philox_randexp(A.seed, chain_id, (uint64_t)t);
  __threadfence();   // the draws are read back by other threads of this workgroup: through L2, not a stale L1 line
  __syncthreads();
  // this thread's draws of the NEXT transition, requested one transition ahead
  double eps_next = 0.0, exp_next = 0.0;
  if (tid < M) eps_next = A.Z_out[zbase + tid];
  if (tid == 0) exp_next = A.lp_out[lbase];
  const int zc = A.o_z, zp = A.o_z + M, red = A.o_red;
  const int lfoutp = (lf.out + 15) & ~15, lfld = (lf.in + 3) & ~3, lfst = (lf.in + 3) >> 2;
  const int SF = A.slot_feats, TM = SF >> 4, ntn = Bp >> 4;

#ifdef SI_CHAIN_STAMPS
  long long cst[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  for (int64_t step = 0; step < A.itr; ++step) {
    // ---- propose (rwmh_propose_kernel): zprop = zcur + sigma_z * eps
    const double eps = eps_next, e_acc = exp_next;
    if (step + 1 < A.itr) {
      if (tid < M) eps_next = A.Z_out[zbase + tid + (int64_t)M * (step + 1)];
      if (tid == 0) exp_next = A.lp_out[lbase + step + 1];
    }
    if (tid < M) chain_lds[zp + tid] = chain_lds[zc + tid] + A.sigma_z * eps;
    __syncthreads();
    SI_CSTAMP(0);
    // ---- K4 (reconstruct_kernel): acc = 0; acc += P[r, m] * z[m]; w = W_swa[r] + acc -- scattered into the padded images
    if (A.p_in_lds) {
      for (int r = tid; r < N; r += CT) {
        double acc = 0.0;
        for (int m = 0; m < M; ++m) acc += chain_lds[A.o_P + r + N * m] * chain_lds[zp + m];
        chain_lds[smap[r]] = chain_lds[A.o_swa + r] + acc;
      }
    } else {
      for (int r = tid; r < N; r += CT) {
        double acc = 0.0;
        for (int m = 0; m < M; ++m) acc += A.P[r + (int64_t)A.ldP * m] * chain_lds[zp + m];
        chain_lds[smap[r]] = chain_lds[A.o_swa + r] + acc;
      }
    }
    __syncthreads();
    SI_CSTAMP(1);
    // ---- stored layers 0 .. L-3 (dense_f64_kernel, plain epilogue)
    int h = A.o_X;   // offset of the current layer input
    for (int l = 0; l < L - 2; ++l) {
      const si_layer& ly = A.lay[l];
      const int o = (l & 1) ? A.o_act1 : A.o_act0;
      const int lout = ly.out, lact = ly.act, outp = (lout + 15) & ~15, ldh = (ly.in + 3) & ~3, nst = (ly.in + 3) >> 2;
      const int ldo = (lout + 3) & ~3;                  // leading dimension of the output = of the next layer's input
      const int ntm = outp >> 4;
      for (int t = wave; t < ntm * ntn; t += CT / 64) {
        const int mt = t % ntm, nt = t / ntm;
        const cd4 acc = chain_tile(A.wp[l], outp, h, ldh, nst, mt, nt, lane);
        const int gi = 16 * mt + c;
        const double bv = chain_lds[A.bp[l] + gi];      // (zero in the padding)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gb = 16 * nt + q + 4 * r;
          if (gi < lout && gb < B) chain_lds[o + gi + ldo * gb] = chain_act(acc[r] + bv, lact);
        }
      }
      __syncthreads();
      h = o;
    }
    SI_CSTAMP(2);
    // ---- layer L-2 with the head folded into its epilogue (dense_f64_kernel<FUSE>): one (slot, batch tile) unit per wave
    for (int u = wave; u < A.fuse_slots * ntn; u += CT / 64) {
      const int sl = u % A.fuse_slots, nt = u / A.fuse_slots;
      cd4 acc[4];
      int gi[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (a >= TM) break;
        const int mt = (sl * SF >> 4) + a;
        gi[a] = 16 * mt + c;
        if (16 * mt < lfoutp) {
          acc[a] = chain_tile(A.wp[L - 2], lfoutp, h, lfld, lfst, mt, nt, lane);
          const double bv = chain_lds[A.bp[L - 2] + gi[a]];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][r] = chain_act(acc[a][r] + bv, lf.act);
        } else {   // a tile of the slot past the padded width: zero weights (what the launch path multiplies by zero)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][r] = chain_act(0.0, lf.act);
        }
      }
      for (int o = 0; o < outL; ++o) {
        double wl[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) wl[a] = (a < TM && gi[a] < lf.out) ? chain_lds[A.wp[L - 1] + o + outL * gi[a]] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double p = 0.0;
#pragma unroll
          for (int a = 0; a < 4; ++a)
            if (a < TM) p = fma(acc[a][r], wl[a], p);
          p += chain_row_ror<8>(p);   // == p += __shfl_xor(p, 8, 16); ... 4, 2, 1 (see chain_row_ror)
          p += chain_row_ror<4>(p);
          p += chain_row_ror<2>(p);
          p += chain_row_ror<1>(p);
          const int gb = 16 * nt + q + 4 * r;
          if (c == 0 && gb < B) chain_lds[A.o_part + (sl * outL + o) * B + gb] = p;
        }
      }
    }
    __syncthreads();
    SI_CSTAMP(3);
    // ---- tail (tail_sse_kernel): 256-thread virtual blocks, one element per thread; four of them side by side
    {
      const int g = tid >> 8, t256 = tid & 255, w4 = wave & 3;
      for (int vb0 = 0; vb0 < A.nblocks; vb0 += CT / 256) {
        const int vb = vb0 + g, idx = vb * 256 + t256;
        double accv = 0.0;
        if (vb < A.nblocks && idx < d) {
          const int o = idx % outL, b = idx / outL;
          double s = 0.0;
          for (int sl = 0; sl < A.fuse_slots; ++sl) s += chain_lds[A.o_part + (sl * outL + o) * B + b];
          double v = s + chain_lds[A.bp[L - 1] + o];
          v = chain_act(v, ll.act);
          const double r = chain_lds[A.o_Y + idx] - v;
          accv += r * r;
        }
        accv = chain_wave_sum(accv);
        if (lane == 0) chain_lds[red + 8 + 4 * g + w4] = accv;
        __syncthreads();
        if (t256 == 0 && vb < A.nblocks)
          chain_lds[A.o_blk + vb] = (chain_lds[red + 8 + 4 * g] + chain_lds[red + 9 + 4 * g]) + (chain_lds[red + 10 + 4 * g] + chain_lds[red + 11 + 4 * g]);
        __syncthreads();
      }
    }
    SI_CSTAMP(4);
    // ---- sse_final_kernel + rwmh_accept_kernel.  With at most 64 block partials ONE wave does both: lanes past the partials
    // add zeros exactly like the idle threads of the 256-thread block, the other three wave sums are exact zeros.
    if (A.nblocks <= 64) {
      if (wave == 0) {
        double acc = lane < A.nblocks ? 0.0 + chain_lds[A.o_blk + lane] : 0.0;
        acc = chain_wave_sum(acc);
        if (lane == 0) {
          const double sse = (acc + 0.0) + (0.0 + 0.0);
          const double lp_new = A.c0 - (sse / A.sigma2) / 2.0;
          const double lp_old = chain_lds[red + 5];
          const bool accept = step == 0 ? true : (-e_acc < lp_new - lp_old);   // NaN compares false => reject, as in Julia
          const double lp_keep = accept ? lp_new : lp_old;
          chain_lds[red + 5] = lp_keep;
          chain_lds[red + 6] = accept ? 1.0 : 0.0;
          A.lp_out[lbase + step] = lp_keep;
          if (accept && step > 0) nacc += 1;
        }
      }
      __syncthreads();
    } else {
      if (tid < 256) {
        double acc = 0.0;
        for (int i = tid; i < A.nblocks; i += 256) acc += chain_lds[A.o_blk + i];
        acc = chain_wave_sum(acc);
        if (lane == 0) chain_lds[red + wave] = acc;
      }
      __syncthreads();
      if (tid == 0) {
        const double sse = (chain_lds[red] + chain_lds[red + 1]) + (chain_lds[red + 2] + chain_lds[red + 3]);
        const double lp_new = A.c0 - (sse / A.sigma2) / 2.0;
        const double lp_old = chain_lds[red + 5];
        const bool accept = step == 0 ? true : (-e_acc < lp_new - lp_old);
        const double lp_keep = accept ? lp_new : lp_old;
        chain_lds[red + 5] = lp_keep;
        chain_lds[red + 6] = accept ? 1.0 : 0.0;
        A.lp_out[lbase + step] = lp_keep;
        if (accept && step > 0) nacc += 1;
      }
      __syncthreads();
    }
    SI_CSTAMP(5);
    if (tid < M) {
      const double zv = chain_lds[red + 6] != 0.0 ? chain_lds[zp + tid] : chain_lds[zc + tid];
      chain_lds[zc + tid] = zv;
      A.Z_out[zbase + tid + (int64_t)M * step] = zv;
    }
    __syncthreads();
    SI_CSTAMP(6);
  }
#ifdef SI_CHAIN_STAMPS
  if (tid == 0 && blockIdx.x == 0 && A.dbg_stamps)
    for (int i = 0; i < 8; ++i) A.dbg_stamps[i] = cst[i];
#endif
  if (tid == 0) A.nacc_out[chain] = nacc;
}

// LDS bytes the kernel needs for this model, or 0 when it does not apply (layout offsets are left in `a`)
size_t chain_loop_plan(ChainLoopArgs& a, size_t lds_limit) {
  const int L = a.L, B = a.B, N = a.N, M = a.M;
  if (L < 2 || L > SI_CHAIN_MAX_LAYERS) return 0;
  auto up = [](int64_t v, int64_t m) { return (v + m - 1) / m * m; };
  auto even = [](int64_t v) { return (v + 1) & ~(int64_t)1; };
  const int64_t Bp = up(B, 16);
  int64_t off = 0, maxld = 4;
  for (int l = 0; l < L; ++l) {   // padded weights + bias per layer (the head keeps its own shape)
    const si_layer& ly = a.lay[l];
    const int64_t outp = l < L - 1 ? up(ly.out, 16) : ly.out, inp = l < L - 1 ? up(ly.in, 4) : ly.in;
    a.wp[l] = (int)off;  off += even(outp * inp);
    a.bp[l] = (int)off;  off += even(outp);
    if (l >= 1 && l <= L - 2) maxld = std::max<int64_t>(maxld, up(ly.in, 4));   // inputs of layers 1 .. L-2 live in the act buffers
  }
  const int64_t d = (int64_t)a.lay[L - 1].out * B;
  a.nblocks = (int)((d + 255) / 256);
  a.o_X = (int)off;       off += even(up(a.lay[0].in, 4) * Bp);
  a.o_act0 = (int)off;    off += even(L > 2 ? maxld * Bp : 0);
  a.o_act1 = (int)off;    off += even(L > 3 ? maxld * Bp : 0);
  a.o_part = (int)off;    off += even((int64_t)a.fuse_slots * d);
  a.o_map = (int)off;     off += even((N + 1) / 2);     // N ints: everything in front of it is zeroed once
  a.o_Y = (int)off;       off += even(d);
  a.o_blk = (int)off;     off += even(a.nblocks);
  a.o_z = (int)off;       off += even(2 * M);
  a.o_red = (int)off;     off += 8 + 16;   // + the wave sums of four virtual blocks side by side
  a.o_swa = (int)off;     off += even(N);
  const int64_t with_p = off + even((int64_t)N * M);
  if ((size_t)with_p * sizeof(double) <= lds_limit) {
    a.p_in_lds = 1;
    a.o_P = (int)off;
    off = with_p;
  } else {
    a.p_in_lds = 0;
    a.o_P = 0;
  }
  if ((size_t)off * sizeof(double) > lds_limit || off > 0x7fffffff / 8) return 0;
  return (size_t)off * sizeof(double);
}

void launch_chain_loop(hipStream_t st, const ChainLoopArgs& a, int nchains, size_t lds) {
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(rwmh_chain_kernel), lds);
  hipLaunchKernelGGL(rwmh_chain_kernel, dim3((unsigned)nchains), dim3(CT), lds, st, a);
}

}  // namespace si
