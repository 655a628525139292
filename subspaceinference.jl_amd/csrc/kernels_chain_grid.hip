// K5 + K6 for NARROW Dense chains (round 5): the model of the reference's only worked example, docs/src/nn_example.md:112-118
// (2-200-50-50-50-1 on 1000 observations, sampled at :188-194), and everything of that class (layer widths <= 256).
//
//   chain_fused_kernel<NB>      K5 of src/space_inference.jl:92-94 in ONE launch: a workgroup owns a batch tile of 16*NB
//                               observations and runs EVERY layer on it.  Activations never leave the CU (LDS images, SURVEY
//                               8(d): "activations stay in LDS/regs when B-tiled"); the weights W_swa + P z' of the chain stream
//                               from L2 straight into the B operand of v_mfma_f64_16x16x4_f64 (lane (q, c) loads
//                               W[16 mt + c][4 s + q]: no staging, no barrier inside a layer); the narrow head and its bias /
//                               activation are applied from the last LDS image; the model outputs yhat go to memory and the
//                               existing fixed-order SSE kernels finish lp.  Chains are stacked in grid.y.
//   rwmh_chain_grid_kernel<NB>  K6 of src/space_inference.jl:111-116 as a PERSISTENT loop over a grid of workgroups: G =
//                               ceil(B / (16 NB)) workgroups per chain, all resident (one per CU).  Per transition every
//                               workgroup proposes the same z' from the same Philox draw, forms ITS rows of W_swa + P z' (K4)
//                               into an L2-resident buffer -> grid barrier -> fused forward on ITS batch tile -> yhat slice ->
//                               grid barrier -> every workgroup sums the squared errors of ALL observations in the fixed order
//                               of sse_partial_kernel / sse_final_kernel and takes the same accept decision.  Two barriers on
//                               an atomic counter per transition instead of seven dependent launches.
//
// SAME BITS as the launch-per-step path (tests/test_gpu_chain_grid.py: array_equal): K4's multiply-then-add per column of P;
// per output element the MFMA k steps of 4 in ascending k (a k step of zeros adds nothing: the 16-deep padding of the big-tile
// kernel is skipped); bias + activation; the head's fma chain over the 16-feature tiles of a feature slot (dense_fused_slot_feats)
// and its 16-lane butterfly; the slots summed in order + bias + activation (tail_sse_kernel); (y - yhat)^2 one element per
// thread, shuffle-down wave sums, (r0 + r1) + (r2 + r3) per 256 elements, the block partials in index order (sse_final_kernel).
// Compiled with -ffp-contract=off: the only fused multiply-adds are the explicit fma of the head and the MFMAs.
//
// Inter-workgroup visibility (per-XCD L2s are not coherent, a CU's L1 is never refreshed): every byte one workgroup hands to
// another -- the weight buffer, the yhat buffer -- is written with agent-scope (sc1, write-through) stores, drained by every
// storing wave (s_waitcnt vmcnt(0)) in front of the workgroup barrier behind which ONE lane adds to the chain's counter, and
// read ONLY with agent-scope (sc1) loads after the polling lane has seen the count and the workgroup has passed a barrier.
// Every spin is bounded by the 100 MHz real-time counter: on a time-out the workgroup raises the status word and leaves; the
// host reports SI_ERR_HIP (never a hang).
#include <algorithm>

#include "chain_common.h"
#include "philox.h"

namespace si {

#ifdef SI_CG_STAMPS   // tools/chain_grid_bench.hip only: cycles per phase of workgroup 0, summed over the transitions
__device__ long long cg_stamp_sum[32];
#endif
#ifdef SI_CG_KNOB   // harness builds only: knock-outs (timing only, results wrong): 1 no W loads, 2 no H reads from LDS, 4 no MFMAs, 8 no image stores
#define SI_CGKNOB(b) ((SI_CG_KNOB) & (b))
#else
#define SI_CGKNOB(b) 0
#endif
#ifdef SI_CG_STAMPS
#define SI_CGSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); cg_stamp_sum[i] += t_ - cg_tlast; cg_tlast = t_; } } while (0)
#define SI_CGSTAMP_DECL long long cg_tlast = __builtin_amdgcn_s_memtime()
#define SI_CGSTAMP_ARG , long long& cg_tlast
#define SI_CGSTAMP_PASS , cg_tlast
#else
#define SI_CGSTAMP(i) do {} while (0)
#define SI_CGSTAMP_DECL do {} while (0)
#define SI_CGSTAMP_ARG
#define SI_CGSTAMP_PASS
#endif

typedef double g4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gdbl;
typedef __attribute__((address_space(1))) unsigned gu32;

extern __shared__ __attribute__((aligned(16))) double cg_lds[];

// a load / store of bytes that another workgroup of the same launch writes / reads: agent scope (global_load / _store ... sc1)
template <bool COH>
__device__ __forceinline__ double cg_ld(const double* p) {
  if constexpr (COH)
    return __hip_atomic_load((gdbl*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    return *p;
}
template <bool COH>
__device__ __forceinline__ void cg_st(double* p, double v) {
  if constexpr (COH)
    __hip_atomic_store((gdbl*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    *p = v;
}

// the activation of the MFMA epilogues (chain_act): identity / relu inline, tanh / sigmoid out of line -- inlined at every
// unrolled store they made the layer loop 12 000 instructions long (same expressions, same library code, same bits)
__device__ __attribute__((noinline)) double cg_act_slow(double v, int act) { return chain_act(v, act); }
__device__ __forceinline__ double cg_act(double v, int act) {
  if (act == SI_ACT_RELU) return v > 0.0 ? v : 0.0;
  if (act == SI_ACT_IDENTITY) return v;
  return cg_act_slow(v, act);
}

constexpr int CG_NT = 256;   // 4 waves, one per SIMD
constexpr int CG_NW = CG_NT / 64;
constexpr int CG_KC = 8;     // k steps per register chunk of W fragments (two chunks in flight)

// ---- the tile program ------------------------------------------------------------------------------------------------
// The matrix layers of a chain are a FLAT list of 16-feature tiles in the order a wave runs them (built on the host once per
// set-up: chain_fused_program): one 64-byte descriptor per tile, read with one scalar load a tile ahead.  The first versions
// walked the layer table inside the kernel (layer loop, tile loop, search for the wave's next tile): ~10 dependent scalar
// loads and a few hundred bookkeeping instructions per tile -- with EVERY load, LDS access and MFMA knocked out the kernels
// took as long as with them (tools/chain_grid_bench.hip knock-outs, profiles/r05_chain_grid_knockouts.log).  A version with
// ONE flat loop over 16-k chunks and a four-chunk register ring was slower still: 170 instructions of state keeping per 4 MFMAs.
// What is kept: a tight k loop inside a tile (addresses advance by a scalar stride), tile-level state from the descriptor.

// The W fragments of one 16-feature tile, as the loads see them: a UNIFORM base (the layer's W, advanced by 4 * out elements
// per k step: scalar registers) + the lane's 32-bit element offset (row + out * q) -- one global_load per fragment and no
// 64-bit vector arithmetic.  The k range is nsf FULL steps (k = 4 st + q < in for every lane) plus, when in % 4 != 0, one
// RAGGED step whose lanes with k >= in must contribute zero: only that one fragment goes through a select, once per tile -- a
// select on every loaded fragment made the compiler wait for each load inside the loop that issued it.
struct CgW {
  const double* W;   // uniform; nullptr: no tile
  uint32_t off;      // per lane: min(16 mt + c, out - 1) + out * q                       (full steps)
  uint32_t off_r;    // per lane: row + out * min(4 nsf + q, in - 1)                       (the ragged step; always a valid element)
  int out4, nsf;     // 4 * out; in / 4
  bool ragged;       // uniform: in % 4 != 0
  bool qok;          // per lane: 4 nsf + q < in
};
__device__ __forceinline__ CgW cg_make(const CgTileD& d, const double* __restrict__ w, int q, int c, bool valid) {
  CgW t;
  const int in = d.in;
  t.W = (valid && !(d.flags & SI_CG_MARKER)) ? w + d.woff : nullptr;
  const int row = d.row0 + c < d.out ? d.row0 + c : d.out - 1;   // rows past `out` only feed outputs that are never used
  t.nsf = in >> 2;
  t.ragged = (in & 3) != 0;
  t.off = (uint32_t)(row + d.out * q);
  const int kr = 4 * t.nsf + q;
  t.qok = kr < in;
  t.off_r = (uint32_t)(row + d.out * (kr < in ? kr : in - 1));
  t.out4 = 4 * d.out;
  return t;
}
// CG_KC full k steps of W fragments: lane (q, c) holds W[row(c)][4 st + q].  No branch, no select: a step past the last full
// one re-reads the last full step (never used).
template <bool COH>
__device__ __forceinline__ void cg_loadw(const CgW& t, int st0, double (&f)[CG_KC]) {
  if (t.nsf > 0) {
#pragma unroll
    for (int s = 0; s < CG_KC; ++s) {
      const int st = st0 + s, last = t.nsf - 1;
      const double* sb = t.W + (int64_t)t.out4 * (st < last ? st : last);   // scalar
      f[s] = SI_CGKNOB(1) ? 0.25 + s : cg_ld<COH>(sb + t.off);
    }
  }
}
template <bool COH>
__device__ __forceinline__ double cg_loadr(const CgW& t) {
  return cg_ld<COH>(t.W + t.off_r);
}
// NS k steps on NB batch sub-tiles: every H fragment is requested from the LDS image first (one latency for the lot), then the
// MFMAs run in ascending k (fragments f[F0 .. F0 + NS) on k steps st0 .. st0 + NS)
template <int NS, int NB, int F0 = 0>
__device__ __forceinline__ void cg_mma(int hb, int ldi, int st0, const double (&f)[CG_KC], g4 (&acc)[NB]) {
  double a[NS][NB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) a[s][nb] = SI_CGKNOB(2) ? 0.5 * s : cg_lds[hb + 4 * (st0 + s) + 16 * ldi * nb];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (SI_CGKNOB(4))
        acc[nb][0] += a[s][nb] * f[F0 + s];
      else
        acc[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][nb], f[F0 + s], acc[nb], 0, 0, 0);
    }
}

// one 16-feature x (16 NB)-observation tile of W * H: acc[nb] holds D[b = 16 nb + q + 4 r][i = 16 mt + c].
// (f0, fr) hold the tile's FIRST chunk of W fragments and its ragged fragment on entry (requested by the tile before it) and
// those of the wave's NEXT tile (`nx`, possibly of the next layer) on return: a weight fragment never waits for an
// activation, so its L2 latency hides behind the epilogue / the layer barrier instead of opening every tile.
// The W fragments of chunk ch + 1 are requested before the MFMAs of chunk ch (straight-line blocks of CG_KC * NB MFMAs: a
// branch per k step makes the compiler move the accumulators between register files at every step); H fragments come from
// the LDS image hs (pitch ldi = 4 j + 2 doubles: the 32 lanes of a ds_read_b64 half hit 32 different bank pairs).
template <int NB, bool COH>
__device__ __forceinline__ void cg_tile(const CgW& t, int hs, int ldi, int lane, double (&f0)[CG_KC], double& fr, g4 (&acc)[NB],
                                        const CgW& nx, const double* __restrict__ bias_nx, double& bvn) {
  const int q = lane >> 4, c = lane & 15;
  const int hb = hs + q + ldi * c;
  const int nfull = t.nsf / CG_KC, rem = t.nsf % CG_KC;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = (g4){0.0, 0.0, 0.0, 0.0};
  double f1[CG_KC];
  const double frv = t.qok ? fr : 0.0;   // (requested a tile ago)
  auto tail = [&](const double (&f)[CG_KC]) {
    const int st0 = CG_KC * nfull;
    switch (rem) {   // (uniform: one scalar branch)
      case 1: cg_mma<1, NB>(hb, ldi, st0, f, acc); break;
      case 2: cg_mma<2, NB>(hb, ldi, st0, f, acc); break;
      case 3: cg_mma<3, NB>(hb, ldi, st0, f, acc); break;
      case 4: cg_mma<4, NB>(hb, ldi, st0, f, acc); break;
      case 5: cg_mma<5, NB>(hb, ldi, st0, f, acc); break;
      case 6: cg_mma<6, NB>(hb, ldi, st0, f, acc); break;
      case 7: cg_mma<7, NB>(hb, ldi, st0, f, acc); break;
      default: break;
    }
    if (t.ragged) {
      double fl[CG_KC];
      fl[0] = frv;
      cg_mma<1, NB>(hb, ldi, t.nsf, fl, acc);
    }
  };
  int ch = 0;
  for (; ch + 1 < nfull; ch += 2) {   // two register sets, no copies: the loads of a chunk have a chunk of MFMAs to land
    cg_loadw<COH>(t, CG_KC * (ch + 1), f1);
    cg_mma<CG_KC, NB>(hb, ldi, CG_KC * ch, f0, acc);
    cg_loadw<COH>(t, CG_KC * (ch + 2), f0);
    cg_mma<CG_KC, NB>(hb, ldi, CG_KC * (ch + 1), f1, acc);
  }
  if (ch < nfull) {
    cg_loadw<COH>(t, CG_KC * (ch + 1), f1);
    cg_mma<CG_KC, NB>(hb, ldi, CG_KC * ch, f0, acc);
    if (nx.W != nullptr) {
      cg_loadw<COH>(nx, 0, f0);
      fr = cg_loadr<COH>(nx);
      bvn = cg_ld<COH>(bias_nx);
    }
    tail(f1);
  } else {
    tail(f0);
    if (nx.W != nullptr) {
      cg_loadw<COH>(nx, 0, f0);
      fr = cg_loadr<COH>(nx);
      bvn = cg_ld<COH>(bias_nx);
    }
  }
}

// The whole chain on one batch tile of 16 NB observations.  The input image of layer 0 (X tile) is image 0; layer l writes
// image 1 + (l & 1).  On return the tile's model outputs are in yhat (global, out_last x B column-major).
// TW = waves that share the batch tile: CG_NW (the 16-feature tiles of a layer dealt over the workgroup's waves, a workgroup
// barrier behind every layer: the shortest critical path, what the persistent loop wants) or 1 (the wave owns its tile: every
// feature tile of every layer, images at `lds0` in a region of its own, NO workgroup barrier anywhere).
template <int NB, int TW, bool COH>
__device__ __forceinline__ void cg_forward(const ChainFusedPlan& p, const double* __restrict__ w, double* __restrict__ yhat,
                                           int64_t b0, int lds0 SI_CGSTAMP_ARG) {
  constexpr int BT = 16 * NB;
  constexpr int NTT = 64 * TW;   // threads that share the tile
  const int tid = TW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = threadIdx.x & 63;
  const int wave = TW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the loops branch uniformly)
  auto tile_sync = [&]() {
    if constexpr (TW == 1)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (LDS is in order per wave: only the compiler's order matters)
    else
      __syncthreads();
  };
  const int q = lane >> 4, c = lane & 15;
  const int L = p.L, B = p.B;
  const si_layer& ll = p.lay[L - 1];
  const int oimg0 = lds0 + p.o_x, oimg1 = lds0 + p.o_buf[0], oimg2 = lds0 + p.o_buf[1];
  auto image = [&](int i) { return i == 0 ? oimg0 : i == 1 ? oimg1 : oimg2; };
  // ---- every layer that runs on the matrix cores: the wave's tile list
  const int pi = TW == 1 ? 0 : 1 + wave;
  const CgTileD* prog = p.prog + p.prog_start[pi];
  const int ntiles = p.prog_count[pi];
  CgTileD d = prog[0];
  CgTileD dn = prog[ntiles > 1 ? 1 : 0];
  CgW cur = cg_make(d, w, q, c, true);
  double f0[CG_KC], fr = 0.0, bv = 0.0;
  auto bias_of = [&](const CgTileD& dd) { return w + dd.boff + (dd.row0 + c < dd.out ? dd.row0 + c : dd.out - 1); };   // (rows past `out` are never used)
  bool pf = cur.W != nullptr;
  if (pf) {
    cg_loadw<COH>(cur, 0, f0);
    fr = cg_loadr<COH>(cur);
    bv = cg_ld<COH>(bias_of(d));
  }
  // the head's operands do not depend on the activations either: requested now, used behind the last layer
  double wl0[4] = {0.0, 0.0, 0.0, 0.0}, blast[SI_FUSE_MAX_OUT] = {0.0, 0.0, 0.0, 0.0};
  if (p.fuse_tail) {
    const int SF = p.slot_feats, outF = p.lay[L - 2].out;
    if (wave < p.fuse_slots * NB) {   // (unconditional loads of clamped, valid addresses: a select on a loaded value makes the compiler wait for it here)
      const int sl = wave % p.fuse_slots;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int gi = sl * SF + 16 * a + c;
        wl0[a] = cg_ld<COH>(w + ll.w_off + (int64_t)ll.out * (gi < outF ? gi : outF - 1));
      }
    }
#pragma unroll
    for (int o = 0; o < SI_FUSE_MAX_OUT; ++o) blast[o] = cg_ld<COH>(w + ll.b_off + (o < ll.out ? o : ll.out - 1));
  }
  for (int t = 0; t < ntiles; ++t) {
    const CgW nx = cg_make(dn, w, q, c, t + 1 < ntiles);
    if (cur.W != nullptr) {
      const int gi = d.row0 + c;
      if (!pf) {   // (the tile behind a marker: nothing was requested for it)
        cg_loadw<COH>(cur, 0, f0);
        fr = cg_loadr<COH>(cur);
        bv = cg_ld<COH>(bias_of(d));
      }
      g4 acc[NB];
      double bvn = 0.0;   // the next tile's bias: requested with its first W fragments, a tile ahead
      cg_tile<NB, COH>(cur, image(d.img_in), d.ldi, lane, f0, fr, acc, nx, bias_of(dn), bvn);
      const int act = d.act & 0xff;
      if (d.flags & SI_CG_GLOBAL) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t gb = b0 + 16 * nb + q + 4 * r;
            if (gi < d.out && gb < B) cg_st<COH>(yhat + gi + (int64_t)d.out * gb, cg_act(acc[nb][r] + bv, act));
          }
      } else if (gi < ((d.out + 3) & ~3) && !SI_CGKNOB(8)) {
        const int ho = image(d.img_out);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r)   // rows out .. ceil4(out) - 1 are the zero k padding of the next layer's image
            cg_lds[ho + gi + d.ldo * (16 * nb + q + 4 * r)] = gi < d.out ? cg_act(acc[nb][r] + bv, act) : 0.0;
      }
      pf = nx.W != nullptr;
      bv = bvn;
    } else {
      pf = false;
    }
    if (d.flags & SI_CG_SYNC) {
      tile_sync();
      SI_CGSTAMP(8 + (d.act >> 8));
    }
    cur = nx;
    d = dn;
    dn = prog[t + 2 < ntiles ? t + 2 : ntiles - 1];   // (a scalar load: waited for a tile later)
  }
  if (p.fuse_tail) {
    // ---- the narrow head on the image of layer L-2 (the epilogue of dense_f64_kernel<FUSE>): per feature slot of SF features
    // p = fma(h_a, wl_a, p) over the slot's 16-feature tiles in order, then the 16-lane butterfly
    const si_layer& lf = p.lay[L - 2];
    const int hs = image(1 + ((L - 2) & 1));
    const int ldh = p.ld[L - 1], SF = p.slot_feats, TM = SF >> 4, outL = ll.out, outF = lf.out;
    const double* Wlast = w + ll.w_off;
    const int opart = lds0 + p.o_part;
    for (int u = wave; u < p.fuse_slots * NB; u += TW) {
      const int sl = u % p.fuse_slots, nb = u / p.fuse_slots;
      for (int o = 0; o < outL; ++o) {
        double wl[4];
        int gi[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          gi[a] = sl * SF + 16 * a + c;
          if (u == wave && o == 0)
            wl[a] = wl0[a];
          else
            wl[a] = cg_ld<COH>(Wlast + o + (int64_t)outL * (gi[a] < outF ? gi[a] : outF - 1));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = 16 * nb + q + 4 * r;
          double pr = 0.0;
#pragma unroll
          for (int a = 0; a < 4; ++a)   // a feature past `out` multiplies a finite activation by zero there: pr unchanged
            if (a < TM && gi[a] < outF) pr = fma(cg_lds[hs + gi[a] + ldh * b], wl[a], pr);
          pr += chain_row_ror<8>(pr);
          pr += chain_row_ror<4>(pr);
          pr += chain_row_ror<2>(pr);
          pr += chain_row_ror<1>(pr);
          if (c == 0) cg_lds[opart + (sl * outL + o) * BT + b] = pr;
        }
      }
    }
    tile_sync();
    SI_CGSTAMP(16);
    // ---- tail_sse_kernel's front half: the slots in order + bias, activation
    for (int e = tid; e < outL * BT; e += NTT) {
      const int o = e % outL, b = e / outL;
      if (b0 + b < B) {
        double s = 0.0;
        for (int sl = 0; sl < p.fuse_slots; ++sl) s += cg_lds[opart + (sl * outL + o) * BT + b];
        const double bl = o == 0 ? blast[0] : o == 1 ? blast[1] : o == 2 ? blast[2] : blast[3];
        cg_st<COH>(yhat + o + (int64_t)outL * (b0 + b), cg_act(s + bl, ll.act));
      }
    }
  }
}

// the X tile as the input image of layer 0: rows k >= in (up to a multiple of 4) and columns b >= B are zero
template <int NB, int TW>
__device__ __forceinline__ void cg_stage_x(const ChainFusedPlan& p, const double* __restrict__ X, int64_t b0, int lds0) {
  constexpr int BT = 16 * NB;
  const int in0 = p.lay[0].in, inp = (in0 + 3) & ~3, ld0 = p.ld[0];
  for (int e = TW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x; e < inp * BT; e += 64 * TW) {
    const int k = e % inp, b = e / inp;
    cg_lds[lds0 + p.o_x + k + ld0 * b] = (k < in0 && b0 + b < p.B) ? X[k + (int64_t)in0 * (b0 + b)] : 0.0;
  }
}

template <int NB, int TW>
__global__ __launch_bounds__(CG_NT) void chain_fused_kernel(ChainFusedPlan p, const double* __restrict__ w, int64_t w_stride,
                                                           const double* __restrict__ X, double* __restrict__ yhat,
                                                           int64_t y_stride) {
  // TW = 1: one WAVE per batch tile of 16 NB observations: the four waves of a workgroup take four neighbouring tiles of the
  // same chain, walk the same weights at about the same time (their fragment loads meet in the CU's L1) and never wait for
  // one another.  TW = CG_NW (chains whose images do not fit four times into a CU's LDS): the workgroup shares one tile.
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t b0 = TW == 1 ? ((int64_t)blockIdx.x * CG_NW + wave) * (16 * NB) : (int64_t)blockIdx.x * (16 * NB);
  if (TW == 1 && b0 >= p.B) return;   // (no workgroup barrier on that path)
  w += (int64_t)blockIdx.y * w_stride;
  yhat += (int64_t)blockIdx.y * y_stride;
  const int lds0 = TW == 1 ? wave * p.lds_doubles : 0;
  SI_CGSTAMP_DECL;
  cg_stage_x<NB, TW>(p, X, b0, lds0);
  if constexpr (TW == 1)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  else
    __syncthreads();
  SI_CGSTAMP(0);
  cg_forward<NB, TW, false>(p, w, yhat, b0, lds0 SI_CGSTAMP_PASS);
  SI_CGSTAMP(17);
}

// ------------------------------------------------------------------------------------------------------------------------------
// host: the LDS plan and the tile program.  Images are [k + ld * b] with ld = ceil4(width) + 2 (conflict-free ds_read_b64
// fragments; rows up to a multiple of 4 exist and are zero).
// Returns the bytes of dynamic LDS for batch tiles of 16 NB observations, or 0 when the chain is not of this class.
size_t chain_fused_plan(ChainFusedPlan& p, const si_layer* layers, int L, int64_t B, int NB, bool fuse_tail, int slot_feats,
                        int fuse_slots) {
  if (L < 1 || L > SI_CHAIN_MAX_LAYERS || B < 1 || B > (1 << 30)) return 0;
  if (fuse_tail && L < 2) return 0;
  const int BT = 16 * NB;
  auto even = [](int64_t v) { return (v + 1) & ~(int64_t)1; };
  p.L = L;
  p.B = (int)B;
  p.fuse_tail = fuse_tail ? 1 : 0;
  p.slot_feats = slot_feats;
  p.fuse_slots = fuse_slots;
  for (int l = 0; l < L; ++l) {
    if (layers[l].kind != SI_LAYER_DENSE || layers[l].act >= SI_ACT_LEAKYRELU || layers[l].in < 1 || layers[l].out < 1) return 0;
    p.lay[l] = layers[l];
    p.ld[l] = ((layers[l].in + 3) & ~3) + 2;
  }
  p.ld[L] = ((layers[L - 1].out + 3) & ~3) + 2;
  if (fuse_tail && (slot_feats < 16 || slot_feats > 64 || slot_feats % 16 != 0 || layers[L - 1].out > SI_FUSE_MAX_OUT)) return 0;
  int64_t off = 0, sz[2] = {0, 0};
  p.o_x = (int)off;
  off += even((int64_t)p.ld[0] * BT);
  for (int l = 0; l + 1 < L; ++l) sz[l & 1] = std::max<int64_t>(sz[l & 1], (int64_t)p.ld[l + 1] * BT);
  p.o_buf[0] = (int)off;
  off += even(sz[0]);
  p.o_buf[1] = (int)off;
  off += even(sz[1]);
  p.o_part = (int)off;
  off += even(fuse_tail ? (int64_t)fuse_slots * layers[L - 1].out * BT : 0);
  p.lds_doubles = (int)off;
  if (off > (int64_t)(160 * 1024) / 8) return 0;
  return (size_t)off * sizeof(double);
}

// The tile program of a chain: list 0 = every tile of every matrix layer in order (one wave per batch tile); lists 1 .. 4 =
// the tiles of wave 0 .. 3 when a workgroup shares the batch tile (tile mt of a layer goes to wave mt % 4; a wave without a
// tile in a layer gets a marker, so that every wave meets every layer barrier).  start / count / chunks: per list.
void chain_fused_program(const si_layer* layers, int L, bool fuse_tail, std::vector<CgTileD>& prog, int start[5], int count[5],
                         int chunks[5]) {
  prog.clear();
  const int nmma = fuse_tail ? L - 1 : L;
  for (int list = 0; list < 5; ++list) {
    start[list] = (int)prog.size();
    int nch = 0;
    for (int l = 0; l < nmma; ++l) {
      const si_layer& ly = layers[l];
      const int ntm = (ly.out + 15) / 16;
      const size_t first = prog.size();
      for (int mt = 0; mt < ntm; ++mt) {
        if (list > 0 && mt % CG_NW != list - 1) continue;
        CgTileD d{};
        d.woff = ly.w_off;
        d.boff = ly.b_off;
        d.out = ly.out;
        d.row0 = 16 * mt;
        d.in = ly.in;
        d.flags = (l == L - 1) ? SI_CG_GLOBAL : 0;   // (only without a fused head: the last layer writes yhat itself)
        d.img_in = l == 0 ? 0 : 1 + ((l - 1) & 1);
        d.ldi = ((ly.in + 3) & ~3) + 2;
        d.img_out = 1 + (l & 1);
        d.ldo = ((ly.out + 3) & ~3) + 2;
        d.act = ly.act | (l << 8);   // (the layer index rides along for the harness's stamps)
        prog.push_back(d);
        nch += (ly.in + 15) / 16;
      }
      if (prog.size() == first) {   // this wave has no tile in layer l
        CgTileD d{};
        d.flags = SI_CG_MARKER;
        d.out = 1;
        d.in = 1;
        d.act = l << 8;
        prog.push_back(d);
        nch += 1;
      }
      prog.back().flags |= SI_CG_SYNC;
    }
    count[list] = (int)prog.size() - start[list];
    chunks[list] = nch;
  }
}

template <int NB, int TW>
static void launch_fused_nb(hipStream_t st, const ChainFusedPlan& p, size_t lds, const double* w, int64_t w_stride, const double* X,
                            double* yhat, int64_t y_stride, int nchains) {
  if (TW == 1) lds *= CG_NW;   // one region of images per wave
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(chain_fused_kernel<NB, TW>), lds);
  const unsigned tiles = (unsigned)((p.B + 16 * NB - 1) / (16 * NB));
  hipLaunchKernelGGL((chain_fused_kernel<NB, TW>), dim3(TW == 1 ? (tiles + CG_NW - 1) / CG_NW : tiles, (unsigned)nchains), dim3(CG_NT), lds, st,
                     p, w, w_stride, X, yhat, y_stride);
}

// wave_tiles: one wave per batch tile (needs CG_NW regions of `lds` bytes) instead of one workgroup per tile
void launch_chain_fused(hipStream_t st, const ChainFusedPlan& p, int NB, bool wave_tiles, size_t lds, const double* w, int64_t w_stride,
                        const double* X, double* yhat, int64_t y_stride, int nchains) {
  if (wave_tiles) {
    switch (NB) {
      case 4: launch_fused_nb<4, 1>(st, p, lds, w, w_stride, X, yhat, y_stride, nchains); break;
      case 2: launch_fused_nb<2, 1>(st, p, lds, w, w_stride, X, yhat, y_stride, nchains); break;
      default: launch_fused_nb<1, 1>(st, p, lds, w, w_stride, X, yhat, y_stride, nchains); break;
    }
  } else {
    switch (NB) {
      case 4: launch_fused_nb<4, CG_NW>(st, p, lds, w, w_stride, X, yhat, y_stride, nchains); break;
      case 2: launch_fused_nb<2, CG_NW>(st, p, lds, w, w_stride, X, yhat, y_stride, nchains); break;
      default: launch_fused_nb<1, CG_NW>(st, p, lds, w, w_stride, X, yhat, y_stride, nchains); break;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// K6: the persistent loop
// ------------------------------------------------------------------------------------------------------------------------------
constexpr unsigned long long CG_TIMEOUT_TICKS = 400000000ull;   // 4 s of the 100 MHz real-time counter per barrier wait

// arrive: every storing wave has drained its sc1 stores; ONE lane adds to the chain's counter
__device__ __forceinline__ void cg_arrive(gu32* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait: ONE lane polls (relaxed agent-scope loads, s_sleep between polls, bounded); the workgroup passes a barrier behind it.
// false = timed out (the status word is raised): the caller must leave the kernel.
__device__ __forceinline__ bool cg_wait(gu32* cnt, unsigned target, gu32* status, int o_flag) {
  int* flag = reinterpret_cast<int*>(cg_lds + o_flag);
  if (threadIdx.x == 0) {
    bool ok = true;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0)
          t0 = now;
        else if (now - t0 > CG_TIMEOUT_TICKS || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          ok = false;
          break;
        }
      }
    }
    if (!ok) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = ok ? 1 : 0;
  }
  asm volatile("" ::: "memory");
  __syncthreads();
  return *flag != 0;
}

template <int NB>
__global__ __launch_bounds__(CG_NT) void rwmh_chain_grid_kernel(ChainGridArgs A) {
  constexpr int BT = 16 * NB;
  const ChainFusedPlan& p = A.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the tile loops branch uniformly)
  const int G = A.G, chain = blockIdx.x / G, g = blockIdx.x % G;
  const int N = A.N, M = A.M, L = p.L, B = p.B;
  const int outL = p.lay[L - 1].out, d = outL * B;
  const int64_t b0 = (int64_t)g * BT;
  double* wbuf = A.wbuf + (int64_t)chain * A.w_stride;
  double* ybuf = A.ybuf + (int64_t)chain * A.y_stride;
  gu32* cnt = (gu32*)(A.cnt + 32 * chain);   // one 128-byte line per chain
  gu32* status = (gu32*)A.status;
  const int zc = A.o_z, zp = A.o_z + M, ze = A.o_z + 2 * M, red = A.o_red;
  // ---- once: this workgroup's X tile, all of Y (every workgroup sums all squared errors), the chain state
  cg_stage_x<NB, CG_NW>(p, A.X, b0, 0);
  if (A.y_in_lds)
    for (int i = tid; i < d; i += CG_NT) cg_lds[A.o_y + i] = A.Y[i];
  for (int m = tid; m < M; m += CG_NT) cg_lds[zc + m] = 0.0;   // rwmh_init_kernel
  const uint32_t chain_id = (uint32_t)(A.chain_id0 + chain);
  const int nblk = (M + 1) >> 1;
  // draws of transition 0 (philox_normal2 / philox_randexp: the functions of rwmh_propose_kernel / rwmh_accept_kernel)
  for (int j = tid; j < nblk; j += CG_NT) {
    double n0, n1;
    philox_normal2(A.seed, chain_id, 0, (uint32_t)j, n0, n1);
    cg_lds[ze + 2 * j] = n0;
    if (2 * j + 1 < M) cg_lds[ze + 2 * j + 1] = n1;
  }
  double lp_cur = -__builtin_inf();
  int64_t nacc = 0;
  unsigned epoch = 0;
  const int64_t zbase = (int64_t)M * A.itr * chain, lbase = A.itr * (int64_t)chain;
  // K4 rows of this workgroup: whole 128-byte lines of the weight buffer
  const int rs = ((N + G - 1) / G + 15) & ~15;
  const int r0 = g * rs, r1 = min(N, r0 + rs);
  __syncthreads();
  SI_CGSTAMP_DECL;

  for (int64_t step = 0; step < A.itr; ++step) {
    // ---- propose (rwmh_propose_kernel): zprop = zcur + sigma_z * eps
    for (int m = tid; m < M; m += CG_NT) cg_lds[zp + m] = cg_lds[zc + m] + A.sigma_z * cg_lds[ze + m];
    __syncthreads();
    SI_CGSTAMP(0);
    // ---- K4 (reconstruct_kernel) on rows [r0, r1): acc = 0; acc += P[r, m] * z[m]; w = W_swa[r] + acc
    for (int r = r0 + tid; r < r1; r += CG_NT) {
      double acc = 0.0;
      const double sv = A.swa[r];
      int m = 0;
      for (; m + 8 <= M; m += 8) {   // eight columns of P requested together, added in column order
        double pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = A.P[r + A.ldP * (m + j)];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += pv[j] * cg_lds[zp + m + j];
      }
      for (; m < M; ++m) acc += A.P[r + A.ldP * m] * cg_lds[zp + m];
      cg_st<true>(wbuf + r, sv + acc);
    }
    SI_CGSTAMP(1);
    cg_arrive(cnt);
    SI_CGSTAMP(2);
    // (while the other workgroups arrive) the draws of the NEXT transition's proposal and of THIS transition's accept test
    if (wave == 1 || M > 128) {
      if (step + 1 < A.itr)
        for (int j = (M > 128 ? tid : lane); j < nblk; j += (M > 128 ? CG_NT : 64)) {
          double n0, n1;
          philox_normal2(A.seed, chain_id, (uint64_t)(step + 1), (uint32_t)j, n0, n1);
          cg_lds[ze + 2 * j] = n0;   // (the proposal of this transition has been formed: eps is free)
          if (2 * j + 1 < M) cg_lds[ze + 2 * j + 1] = n1;
        }
    }
    if (tid == 128 && step > 0) cg_lds[red + 7] = philox_randexp(A.seed, chain_id, (uint64_t)step);   // (wave 0 is about to poll)
    epoch += 1;
    if (!cg_wait(cnt, epoch * (unsigned)G, status, A.o_flag)) return;
    SI_CGSTAMP(3);
    // ---- K5 on this workgroup's batch tile, weights from the buffer every workgroup of the chain has written
    cg_forward<NB, CG_NW, true>(p, wbuf, ybuf, b0, 0 SI_CGSTAMP_PASS);
    SI_CGSTAMP(4);
    cg_arrive(cnt);
    epoch += 1;
    if (!cg_wait(cnt, epoch * (unsigned)G, status, A.o_flag)) return;
    SI_CGSTAMP(5);
    // ---- (y - yhat)^2 over ALL observations in the order of sse_partial_kernel (256 elements per block, one per thread) ...
    for (int vb0 = 0; vb0 < A.nblocks; vb0 += 4) {
      double ev[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // (requested together: one trip to the L2 / the fabric for four blocks)
        const int idx = (vb0 + j) * 256 + tid;
        ev[j] = idx < d ? cg_ld<true>(ybuf + idx) : 0.0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = (vb0 + j) * 256 + tid;
        double accv = 0.0;
        if (idx < d) {
          double yv;
          if (A.y_in_lds)
            yv = cg_lds[A.o_y + idx];
          else
            yv = A.Y[idx];
          const double r = yv - ev[j];
          accv += r * r;
        }
        accv = chain_wave_sum(accv);
        if (lane == 0) cg_lds[red + 8 + 4 * j + wave] = accv;
      }
      __syncthreads();
      if (tid < 4 && vb0 + tid < A.nblocks) {
        const int rb = red + 8 + 4 * tid;
        cg_lds[A.o_blk + vb0 + tid] = (cg_lds[rb] + cg_lds[rb + 1]) + (cg_lds[rb + 2] + cg_lds[rb + 3]);
      }
      __syncthreads();
    }
    // ---- ... and of sse_final_kernel, then rwmh_accept_kernel
    {
      double acc = 0.0;
      for (int i = tid; i < A.nblocks; i += 256) acc += cg_lds[A.o_blk + i];
      acc = chain_wave_sum(acc);
      if (lane == 0) cg_lds[red + wave] = acc;
    }
    __syncthreads();
    if (tid == 0) {
      const double sse = (cg_lds[red] + cg_lds[red + 1]) + (cg_lds[red + 2] + cg_lds[red + 3]);
      const double lp_new = A.c0 - (sse / A.sigma2) / 2.0;
      const double e_acc = cg_lds[red + 7];
      const bool accept = step == 0 ? true : (-e_acc < lp_new - lp_cur);   // NaN compares false => reject, as in Julia
      lp_cur = accept ? lp_new : lp_cur;
      cg_lds[red + 6] = accept ? 1.0 : 0.0;
      if (accept && step > 0) nacc += 1;
      if (g == 0) A.lp_out[lbase + step] = lp_cur;
    }
    __syncthreads();
    {
      const bool acc = cg_lds[red + 6] != 0.0;
      for (int m = tid; m < M; m += CG_NT) {
        const double zv = acc ? cg_lds[zp + m] : cg_lds[zc + m];
        cg_lds[zc + m] = zv;
        if (g == 0) A.Z_out[zbase + m + (int64_t)M * step] = zv;
      }
    }
    __syncthreads();
    SI_CGSTAMP(6);
  }
  if (tid == 0 && g == 0) A.nacc_out[chain] = nacc;
}

// LDS layout of the loop behind the images of the fused forward; returns bytes (0: does not apply)
size_t chain_grid_plan(ChainGridArgs& a, size_t lds_fused) {
  if (lds_fused == 0) return 0;
  auto even = [](int64_t v) { return (v + 1) & ~(int64_t)1; };
  const int64_t d = (int64_t)a.p.lay[a.p.L - 1].out * a.p.B;
  int64_t off = (int64_t)(lds_fused / sizeof(double));
  a.y_in_lds = d <= 4096 ? 1 : 0;
  a.o_y = (int)off;
  off += even(a.y_in_lds ? d : 0);
  a.o_blk = (int)off;
  off += even(a.nblocks);
  a.o_z = (int)off;
  off += even(3 * (int64_t)a.M);
  a.o_red = (int)off;
  off += 24;
  a.o_flag = (int)off;
  off += 2;
  if (off > (int64_t)(160 * 1024 - 256) / 8) return 0;
  return (size_t)off * sizeof(double);
}

template <int NB>
static hipError_t launch_grid_nb(hipStream_t st, const ChainGridArgs& a, int nchains, size_t lds) {
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(rwmh_chain_grid_kernel<NB>), lds);
  hipLaunchKernelGGL(rwmh_chain_grid_kernel<NB>, dim3((unsigned)(a.G * nchains)), dim3(CG_NT), lds, st, a);
  return hipGetLastError();
}

// `lds` is raised to more than half of a CU's LDS so that two workgroups of the loop never share a CU: the hand-off forms used
// above are the ones measured at one workgroup per CU, and a resident grid needs nchains * G <= number of CUs anyway
hipError_t launch_chain_grid(hipStream_t st, const ChainGridArgs& a, int NB, int nchains, size_t lds) {
  lds = std::max(lds, (size_t)(81 * 1024));
  switch (NB) {
    case 4: return launch_grid_nb<4>(st, a, nchains, lds);
    case 2: return launch_grid_nb<2>(st, a, nchains, lds);
    default: return launch_grid_nb<1>(st, a, nchains, lds);
  }
}

}  // namespace si
