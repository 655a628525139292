// Conv / MaxPool / flatten layers of a Flux Chain in fp64: forward and reverse sweep.
//
// The reference restructures ANY Chain (`model_re`, src/libs.jl:55-57) and evaluates it on the full data
// (src/space_inference.jl:94); BASELINE config 4 is a Conv CNN.  Flux 0.11.2 / NNlib 0.7.23 semantics restated in
// oracle/subspace_oracle.py (conv_forward: TRUE convolution, i.e. flipped kernel; weight (KW, KH, CIN, COUT) column-major
// then bias; WHCN activations; MaxPool(k; stride = k, pad = 0)).
//
// Design.  Inside a convolutional stack activations live CHANNEL-FASTEST ("CWHN": element (c, w, h, n) at
// c + Cp*(w + W*(h + H*n)), Cp = channels rounded up to even), so that
//   * a convolution is the Dense GEMM of kernels_gemm / gemm_pipeline.h applied to pixels:  Out[co, pos] =
//     act(sum_k' Wp[co, k'] * patch[k', pos] + b[co]),  k' = cin + Cp*(a + KW*c),  pos = wo + Wo*(ho + Ho*n) -- the output
//     [COUTp x positions] IS the CWHN tensor of the next layer;
//   * the im2col operand is never materialised: the B-operand stager of the GEMM pipeline gathers patch[k', pos] from
//     the input tensor on its way to LDS (16-byte loads along cin; padding / out-of-range taps become zeros);
//   * the weights are re-packed per evaluation (they change with every z) by a tiny kernel into Wp[COUTp x Kp]
//     (kernel flipped, pad channels zero, Kp = K' rounded up to 16) -- 1 M weights, microseconds;
//   * MaxPool and the layout changes at the ends of the stack (WHCN input -> CWHN once at set-up; CWHN -> the
//     reference's WHC feature order at `flatten`) are HBM-bound index kernels.
// Reverse sweep: dX is the same gather-GEMM on the Delta tensor with the transposed, un-flipped weights (fractional
// stride handled by a divisibility test in the gather); dW is a split-K GEMM over positions whose B operand is the
// transposed im2col gather; db is the row sum of Delta.
#include <algorithm>

#include "gemm_pipeline.h"

namespace si {

__device__ __forceinline__ real conv_act(real v, int act) {
  switch (act) {
    case SI_ACT_RELU: return v > 0.0 ? v : 0.0;
    case SI_ACT_TANH: return tanh(v);
    case SI_ACT_SIGMOID: return 1.0 / (1.0 + exp(-v));
    default: return v;   // (GEMM epilogues only see the first four: launch_conv_forward finishes the later ones elementwise)
  }
}
__device__ __forceinline__ real conv_dact(real h, int act) {
  switch (act) {
    case SI_ACT_RELU: return h > 0.0 ? 1.0 : 0.0;
    case SI_ACT_TANH: return 1.0 - h * h;
    case SI_ACT_SIGMOID: return h * (1.0 - h);
    default: return 1.0;
  }
}

static int conv_pick_bm(int rows) {
  if (rows <= 64) return 64;
  auto padded = [&](int bm) { return (rows + bm - 1) / bm * bm; };
  const int p96 = padded(96), p128 = padded(128), p64 = padded(64);
  if (p96 < p128 && p96 <= p64) return 96;
  return p128 <= p64 ? 128 : 64;
}
static unsigned idx_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  return (unsigned)(b < 1 ? 1 : b);
}

// `y ≈ x` of NNlib's ∇maxpool (Julia isapprox: rtol = sqrt(eps), atol = 0), see pool_chosen below
__device__ __forceinline__ bool pool_approx(real x, real y) {
  return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= 1.4901161193847656e-08 * fmax(fabs(x), fabs(y)));
}

// ------------------------------------------------------------------------------------------------ gather stagers
// B operand of the forward / data-gradient GEMM: B(k', pos) = T[cin + Cp*(wi + Wi*(hi + Hi*img))] with
//   (wi, hi) = ((wo, ho) * snum - pad + (a, c) * dil) / sden   (zero when out of range or not divisible by sden).
// LDS image "k-fast" [pos][18] like Stager<.., LAY = 1>.  Each thread owns NREG (position, k-pair) slots; the position
// part of the address is fixed per slot, the tap part changes per k tile -- block-uniform when Cp % 16 == 0 (CELLU).
// POOLP: the position index runs over 2 x 2 / stride-2 pooling windows, four windows per aligned slice of 16 positions,
// input-major inside the slice -- pos = 16 * slice + (window & 3) + 4 * (dx + 2 dy) -- so that the four inputs of a window
// are the four accumulator registers of ONE lane: the pooling epilogue needs no cross-lane traffic (conv_gemm_pool_kernel).
template <int R, int NT, bool CELLU, bool DEN, bool POOLP = false>
struct GatherK {
  static constexpr int LAY = 1;
  static constexpr int NREG = (16 * R + NT * 2 - 1) / (NT * 2);
  static constexpr int RP = R + 16, KP = 18;
  static constexpr int LDS_ELEMS = R * KP;
  // every slot is live and slot r sits NT / 8 positions behind slot 0 at the same k pair (keeps the per-slot state small:
  // the 128 x 128 kernels run at a 128-register budget and spilled into their k loop with per-slot copies of these)
  static_assert((16 * R) % (NT * 2) == 0 && (NT * 2) % 16 == 0, "GatherK: R * 16 must be a multiple of 2 NT");
  int pb[NREG], wb[NREG], hb[NREG];
  int kk, lds0;
  int t_cin, t_adil, t_cdil;   // CELLU: the (block-uniform) tap of the NEXT k tile; the mainloop asks for the tiles in order
  real2 reg[NREG];
  const real* T;
  ConvGeom g;

  __device__ __forceinline__ void init(const real* T_, const ConvGeom& g_, int64_t n0, int64_t npos, int tid) {
    T = T_;
    g = g_;
    const int wh = g.Wo * g.Ho;
    kk = (tid * 2) & 15;
    lds0 = (tid >> 3) * KP + kk;
    t_cin = 0;
    t_adil = 0;
    t_cdil = 0;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      const int rr = (tid >> 3) + (NT / 8) * r;
      int64_t pos = n0 + rr;
      int img, ho, wo;
      if constexpr (POOLP) {
        // inside every aligned slice of 16 positions: position j = (window j & 3) + 4 * (input j >> 2) -- the four inputs of a
        // window then are the four accumulator REGISTERS of one lane (D[n = q + 4r][m = c]: window q, input r).
        // The clamp is on the WINDOW: in the last slice of a tensor whose window count is not a multiple of 4 the inputs
        // 1..3 of the valid windows sit at positions >= npos (clamping the position there handed them the first input of a
        // window past the end -- wrong maxima in the last windows, and a read behind the input tensor).
        int64_t win = ((pos & ~(int64_t)15) >> 2) + SI_GP_PWIN((int)(pos & 15));
        const int64_t nwin = npos >> 2;
        if (win > nwin - 1) win = nwin - 1;  // clamped windows only feed outputs that are never stored
        const int e = SI_GP_PIN((int)(pos & 15)), W2 = g.Wo >> 1, wh2 = W2 * (g.Ho >> 1);
        img = (int)(win / wh2);
        const int sp = (int)(win - (int64_t)img * wh2);
        const int ho2 = sp / W2, wo2 = sp - ho2 * W2;
        wo = 2 * wo2 + (e & 1);
        ho = 2 * ho2 + (e >> 1);
      } else {
        if (pos > npos - 1) pos = npos - 1;  // clamped positions only feed outputs that are never stored
        img = (int)(pos / wh);
        const int sp = (int)(pos - (int64_t)img * wh);
        ho = sp / g.Wo;
        wo = sp - ho * g.Wo;
      }
      pb[r] = img * (int)g.img_stride;
      wb[r] = wo * g.snum_w - g.pad_w;
      hb[r] = ho * g.snum_h - g.pad_h;
    }
  }
  __device__ __forceinline__ void fetch(int r, int a_dil, int c_dil, int cin, bool kvalid) {
    int wi = wb[r] + a_dil, hi = hb[r] + c_dil;
    bool ok = kvalid && wi >= 0 && hi >= 0;
    if constexpr (DEN) {
      ok = ok && (wi % g.sden_w == 0) && (hi % g.sden_h == 0);
      wi /= g.sden_w;
      hi /= g.sden_h;
    }
    ok = ok && wi < g.Wi && hi < g.Hi;
    const int off = ok ? pb[r] + g.Cp * (wi + g.Wi * hi) + cin : 0;
    const real2 v = *reinterpret_cast<const real2*>(T + off);
    reg[r] = ok ? v : SI_GP_MAKE2(0.0, 0.0);
  }
  __device__ __forceinline__ void load(int kt) {
    if constexpr (CELLU) {
      // Cp % 16 == 0: a k tile lies inside one tap; the tap advances by a carry instead of two divisions per tile
#pragma unroll
      for (int r = 0; r < NREG; ++r) fetch(r, t_adil, t_cdil, t_cin + kk, true);
      t_cin += 16;
      if (t_cin >= g.Cp) {
        t_cin = 0;
        t_adil += g.dil_w;
        if (t_adil >= g.KW * g.dil_w) {
          t_adil = 0;
          t_cdil += g.dil_h;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < NREG; ++r) {
        const int kp = 16 * kt + kk;
        const int cell = kp / g.Cp, cin = kp - cell * g.Cp;
        const int c = cell / g.KW, a = cell - c * g.KW;
        fetch(r, a * g.dil_w, c * g.dil_h, cin, kp < g.Kvalid);
      }
    }
  }
  __device__ __forceinline__ void store(real* dst) const {
#pragma unroll
    for (int r = 0; r < NREG; ++r) *reinterpret_cast<real2*>(dst + lds0 + r * (NT / 8) * KP) = reg[r];
  }
  // k is padded to whole tiles (Kp % 16 == 0, taps past Kvalid read as zero): there is no ragged tile
  __device__ __forceinline__ void load_edge(int kt, int64_t) { load(kt); }
  __device__ __forceinline__ void store_edge(real* dst, int, int64_t) const { store(dst); }
};

// B operand of the weight-gradient GEMM: B(k = pos, n = k') = patch[k', pos] -- rows are the taps (fixed per slot), the
// k index walks over positions (split-K).  LDS image "row-fast" [pos][R + 16] like Stager<.., LAY = 0>.
template <int R, int NT>
struct GatherN {
  static constexpr int LAY = 0;
  static constexpr int NREG = (16 * R + NT * 2 - 1) / (NT * 2);
  static constexpr int RP = R + 16, KP = 18;
  static constexpr int LDS_ELEMS = 16 * RP;
  // every slot is live; all slots of a thread share the tap row rr = (2 tid) % R and sit (2 NT / R) positions apart
  static_assert((16 * R) % (NT * 2) == 0 && (NT * 2) % R == 0, "GatherN: 2 NT must be a multiple of R and divide 16 R");
  static constexpr int KSTEP = NT * 2 / R;
  // per slot: the output pixel (wo, ho, image offset + channel) of the CURRENT k tile.  The mainloop asks for the k tiles
  // in order, once each, so the pixel is advanced by 16 positions per load() with a mixed-radix carry -- the 64-bit
  // division pos -> (image, ho, wo) per slot and tile made this kernel VALU-bound (half the speed of the forward
  // convolution) when it was recomputed every time.  left = positions of the slot's column that remain in the split
  // (0 for a tap row past Kvalid).
  // Only slot 0's pixel is kept; slot r's is KSTEP * r positions further (a carry per digit, like the advance per tile).
  int aoff, coff, lds0;
  int left0, wo0, ho0, ioff0;
  real2 reg[NREG];
  const real* T;
  ConvGeom g;
  int dw16, dh16, di16, dws, dhs, dis;

  __device__ __forceinline__ void init(const real* T_, const ConvGeom& g_, int64_t n0, int64_t k0, int64_t kend, int tid) {
    static_assert(KSTEP <= 16, "slots of a thread lie inside one k tile");
    T = T_;
    g = g_;
    const int wh = g.Wo * g.Ho;
    dw16 = 16 % g.Wo;
    dh16 = (16 / g.Wo) % g.Ho;
    di16 = (16 / wh) * (int)g.img_stride;
    dws = KSTEP % g.Wo;
    dhs = (KSTEP / g.Wo) % g.Ho;
    dis = (KSTEP / wh) * (int)g.img_stride;
    const int rr = (tid * 2) % R, kq0 = (tid * 2) / R;
    const int kp = (int)n0 + rr;
    const int cell = kp / g.Cp;
    const int c = cell / g.KW, a = cell - c * g.KW;
    aoff = a * g.dil_w - g.pad_w;
    coff = c * g.dil_h - g.pad_h;
    lds0 = kq0 * RP + rr;
    const int64_t pos = k0 + kq0;
    const int img = (int)(pos / wh), sp = (int)(pos - (int64_t)img * wh);
    ho0 = sp / g.Wo;
    wo0 = sp - ho0 * g.Wo;
    ioff0 = img * (int)g.img_stride + (kp - cell * g.Cp);
    const int64_t l = kend - pos;
    left0 = (l > 0 && kp < g.Kvalid) ? (int)(l < 0x7fffffff ? l : 0x7fffffff) : 0;
  }
  __device__ __forceinline__ static void step(int& w, int& h, int& im, int dw, int dh, int di, const ConvGeom& g) {
    w += dw;   // each digit overflows at most once (w + dw < 2 Wo, h + dh + 1 < 2 Ho)
    h += dh;
    im += di;
    if (w >= g.Wo) {
      w -= g.Wo;
      ++h;
    }
    if (h >= g.Ho) {
      h -= g.Ho;
      im += (int)g.img_stride;
    }
  }
  __device__ __forceinline__ void load(int kt) {
    int w = wo0, h = ho0, im = ioff0;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      const int wi = w * g.snum_w + aoff, hi = h * g.snum_h + coff;
      const bool ok = 16 * kt + KSTEP * r < left0 && wi >= 0 && hi >= 0 && wi < g.Wi && hi < g.Hi;
      const int off = ok ? im + g.Cp * (wi + g.Wi * hi) : 0;
      const real2 v = *reinterpret_cast<const real2*>(T + off);
      reg[r] = ok ? v : SI_GP_MAKE2(0.0, 0.0);
      if (r + 1 < NREG) step(w, h, im, dws, dhs, dis, g);
    }
    step(wo0, ho0, ioff0, dw16, dh16, di16, g);   // the next k tile: 16 positions further
  }
  __device__ __forceinline__ void store(real* dst) const {
#pragma unroll
    for (int r = 0; r < NREG; ++r) *reinterpret_cast<real2*>(dst + lds0 + r * KSTEP * RP) = reg[r];
  }
  __device__ __forceinline__ void load_edge(int kt, int64_t) { load(kt); }  // positions past the split's end are masked in load()
  __device__ __forceinline__ void store_edge(real* dst, int, int64_t) const { store(dst); }
};

// ------------------------------------------------------------------------------------------------ GEMM kernels
// Out[m + Mp*pos] = f(sum_k' Wp[m + Mp*k'] * gather(k', pos))      (forward: f = act(. + bias[m]); data gradient: f = id)
template <int BM, int BN, int WM, int WN, int MINW, bool CELLU, bool DEN, bool BIASACT>
__global__ __launch_bounds__(64 * WM * WN, MINW) void conv_gemm_kernel(const real* __restrict__ Wp, int Mp,
                                                                       const real* __restrict__ T, real* __restrict__ Out,
                                                                       const real* __restrict__ bias, ConvGeom g, int64_t npos,
                                                                       int Kp, int act, int nMt, int64_t nNt) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  using SA = Stager<BM, 0, NT, true>;
  using SB = GatherK<BN, NT, CELLU, DEN>;
  extern __shared__ real smem[];
  const int64_t bid = blockIdx.x;
  int mt;
  int64_t nt;
  if (nNt >= 8) {  // XCD-grouped: the row tiles of one position panel share an XCD's L2 (the gathered patches overlap)
    const int xcd = (int)(bid & 7);
    const int64_t j = bid >> 3;
    mt = (int)(j % nMt);
    nt = (j / nMt) * 8 + xcd;
  } else {
    mt = (int)(bid % nMt);
    nt = bid / nMt;
  }
  if (nt >= nNt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int m0 = mt * BM;
  const int64_t n0 = nt * BN;
  r4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (r4){0.0, 0.0, 0.0, 0.0};
  SA sa;
  SB sb;
  sa.init(Wp, Mp, m0, Mp, 0, tid);
  sb.init(T, g, n0, npos, tid);
  gemm_mainloop<BM, BN, WM, WN, true>(sa, sb, smem, Kp / 16, (int64_t)Kp, wm, wn, lane, acc, 0);
  gemm_epilogue<BM, BN, WM, WN, true, 2 * (SA::LDS_ELEMS + SB::LDS_ELEMS)>(
      acc, smem, Out, (int64_t)Mp, m0, n0, Mp, npos, wm, wn, lane, wave, [&](real v, int64_t, int gm) {
        if constexpr (BIASACT) v = conv_act(v + bias[gm], act);
        return v;
      });
}

// Forward convolution + bias + activation + MaxPool((2, 2)) in one kernel (sampling path: the un-pooled activation is
// never needed).  Positions run input-major inside every slice of 16 (GatherK<POOLP>), so in the MFMA output
// D[n = q + 4r][m = c] the four inputs of pooling window q of the slice are the four accumulator registers r of one lane:
// three fmax, no cross-lane traffic (round 2 had them in four lane groups: two ds_bpermute pairs per value), and lane group
// q stores window q -- Out[m + Mp * window] is the pooled CWHN tensor (4x fewer bytes written,
// and the MaxPool pass with its read of the full activation disappears: 2.1 + 2.7 GB at the first layer of the cfg4 CNN).
// IDX (gradient mode): additionally stores, per pooled element, WHICH of the four window inputs the reverse sweep routes
// the gradient to -- the first one (dx + 2 dy order = NNlib's kw-fastest scan) that is ≈ the maximum -- as one byte.
// With it the un-pooled activation never has to exist: act'(chosen input) = act'(pooled output), so the Delta tensor
// of the layer follows from (pooled gradient, pooled output, index) alone (pool2_bwd_idx_kernel).
template <int BM, int BN, int WM, int WN, int MINW, bool CELLU, bool IDX>
__global__ __launch_bounds__(64 * WM * WN, MINW) void conv_gemm_pool_kernel(const real* __restrict__ Wp, int Mp,
                                                                            const real* __restrict__ T, real* __restrict__ Out,
                                                                            const real* __restrict__ bias, ConvGeom g, int64_t npos,
                                                                            int Kp, int act, int nMt, int64_t nNt,
                                                                            uint8_t* __restrict__ Idx) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  using SA = Stager<BM, 0, NT, true>;
  using SB = GatherK<BN, NT, CELLU, false, true>;
  extern __shared__ real smem[];
  const int64_t bid = blockIdx.x;
  int mt;
  int64_t nt;
  if (nNt >= 8) {
    const int xcd = (int)(bid & 7);
    const int64_t j = bid >> 3;
    mt = (int)(j % nMt);
    nt = (j / nMt) * 8 + xcd;
  } else {
    mt = (int)(bid % nMt);
    nt = bid / nMt;
  }
  if (nt >= nNt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int m0 = mt * BM;
  const int64_t n0 = nt * BN;
  r4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (r4){0.0, 0.0, 0.0, 0.0};
  SA sa;
  SB sb;
  sa.init(Wp, Mp, m0, Mp, 0, tid);
  sb.init(T, g, n0, npos, tid);
  gemm_mainloop<BM, BN, WM, WN, true>(sa, sb, smem, Kp / 16, (int64_t)Kp, wm, wn, lane, acc, 0);
  const int q = lane >> 4, c = lane & 15;
  const int64_t nwin = npos >> 2;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int gm = m0 + wm * (BM / WM) + a * 16 + c;
    const real bv = gm < Mp ? bias[gm] : 0.0;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      // D[n = q + 4r][m = c]: window q of the slice, its input r (dx + 2 dy: NNlib's kw-fastest scan order)
      real vin[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) vin[r] = conv_act(acc[a][b][r] + bv, act);
      const real o = fmax(fmax(vin[0], vin[1]), fmax(vin[2], vin[3]));
      int oi = 4;
      if constexpr (IDX) {
#pragma unroll
        for (int r = 3; r >= 0; --r) oi = pool_approx(o, vin[r]) ? r : oi;   // the FIRST input that is ≈ the maximum
      }
      const int64_t win = ((n0 + wn * (BN / WN) + b * 16) >> 2) + q;
      if (gm < Mp && win < nwin) {
        Out[gm + (int64_t)Mp * win] = o;
        if constexpr (IDX) Idx[gm + (int64_t)Mp * win] = (uint8_t)oi;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ first layers
// A FIRST conv layer (few input channels: Kvalid = Cp*KW*KH <= 36 taps, COUTp <= 64) fused with MaxPool((2, 2)).  In the
// GEMM pipeline above such a layer is all prologue and epilogue: 3 k tiles per 64 x 128 output tile, 32 768 workgroups
// at the cfg4 CNN, 0.72 ms against 0.25 ms of matrix-pipe time.  Here a WAVE owns 16 positions x all output channels:
//   * the packed weights (<= 41 KB) sit in LDS for the whole kernel ([k'][COUT16 + 16]: conflict-free operand reads);
//   * the patch operand needs NO staging at all: k' = cin + Cp*tap, so MFMA k step s, lane (q, c) wants input channel /
//     tap (4s + q) of position c -- ONE 8-byte global load per lane and k step, coalesced over the 4 channels of a pixel
//     and the neighbouring pixels, the next slice's fragments in flight under this slice's MFMAs;
//   * no barrier after the weights are in; waves walk the position slices grid-stride (input-major inside a slice, so the
//     pooling epilogue is lane-local like conv_gemm_pool_kernel's).
// The k steps run in the same order with the same 4-wide grouping as the GEMM pipeline and skipped steps only ever added
// zeros, so the results are bit-identical to conv_gemm_pool_kernel's.
template <int NS, int NTM, bool IDX>
__global__ __launch_bounds__(256, 3) void conv_first_pool_kernel(const real* __restrict__ Wp, int Mp, const real* __restrict__ T,
                                                              real* __restrict__ Out, const real* __restrict__ bias, ConvGeom g,
                                                              int64_t npos, int Kp, int act, uint8_t* __restrict__ Idx) {
  constexpr int MP = NTM * 16 + 16;   // LDS row pitch
  extern __shared__ real sW[];      // [4 * NS][MP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, c = lane & 15;
  for (int e = tid; e < 4 * NS * MP; e += 256) {
    const int k = e / MP, m = e - k * MP;
    sW[e] = (k < Kp && m < Mp) ? Wp[m + (int64_t)Mp * k] : 0.0;
  }
  // tap (a, c) and input channel of this lane's k index in every k step: position-independent, packed into one register
  // per k step (offsets biased by 128; bit 31 = a padded k index, reads as zero)
  unsigned tp[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int kp = 4 * s + q;
    const int tap = kp / g.Cp, cc = tap / g.KW;
    const int aw = (tap - cc * g.KW) * g.dil_w - g.pad_w + 128, ch = cc * g.dil_h - g.pad_h + 128;
    tp[s] = (unsigned)(kp - tap * g.Cp) | ((unsigned)(aw & 0xff) << 8) | ((unsigned)(ch & 0xff) << 16) |
            (kp < g.Kvalid ? 0u : 0x80000000u);
  }
  __syncthreads();
  const int W2 = g.Wo >> 1, wh2 = W2 * (g.Ho >> 1);
  const int64_t nslice = (npos + 15) >> 4, nwin = npos >> 2;
  const int64_t sstride = (int64_t)gridDim.x * 4;
  auto gather = [&](int64_t sl, real(&f)[NS]) {
    // lane c of the patch operand = position c of the slice = window (c & 3), input (c >> 2) of it (see GatherK<POOLP>);
    // 32-bit index arithmetic: the caller guarantees fewer than 2^31 positions
    int win = 4 * (int)sl + SI_GP_PWIN(c);
    if (win > (int)nwin - 1) win = (int)nwin - 1;   // clamped windows are never stored
    const int e = SI_GP_PIN(c);
    const int img = win / wh2, sp = win - img * wh2;
    const int ho2 = sp / W2, wo2 = sp - ho2 * W2;
    const int wb = (2 * wo2 + (e & 1)) * g.snum_w, hb = (2 * ho2 + (e >> 1)) * g.snum_h;
    const real* base = T + (int64_t)img * g.img_stride;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int wi = wb + (int)((tp[s] >> 8) & 0xff) - 128, hi = hb + (int)((tp[s] >> 16) & 0xff) - 128;
      const bool ok = (int)tp[s] >= 0 && wi >= 0 && hi >= 0 && wi < g.Wi && hi < g.Hi;
      const real v = base[ok ? g.Cp * (wi + g.Wi * hi) + (int)(tp[s] & 0xff) : 0];
      f[s] = ok ? v : 0.0;
    }
  };
  real bv[NTM];
#pragma unroll
  for (int t = 0; t < NTM; ++t) bv[t] = 16 * t + c < Mp ? bias[16 * t + c] : 0.0;
  const real* wfrag = sW + q * MP + c;
  int64_t sl = (int64_t)blockIdx.x * 4 + wave;
  real fcur[NS], fnext[NS];
  if (sl < nslice) gather(sl, fcur);
  for (; sl < nslice; sl += sstride) {
    const bool more = sl + sstride < nslice;
    if (more) gather(sl + sstride, fnext);   // in flight under the MFMAs below
    r4 acc[NTM];
#pragma unroll
    for (int t = 0; t < NTM; ++t) acc[t] = (r4){0.0, 0.0, 0.0, 0.0};
    // the weight fragments are re-read from LDS for every slice (4 cycles of LDS time per MFMA of 64): kept in registers --
    // where the compiler puts loop-invariant loads by itself -- they cost 8 registers per k step and channel tile and
    // the kernel ran at two waves per SIMD instead of three (measured: 6.52 against 6.46 ms of conv time per transition)
    int woff = 0;
    asm volatile("" : "+v"(woff));
    const real* wf = wfrag + woff;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int t = 0; t < NTM; ++t)
        acc[t] = SI_GP_MFMA(fcur[s], wf[4 * s * MP + 16 * t], acc[t]);
    }
    // D[position q + 4r][channel c]: window q of the slice, input r of the window -- the maximum is lane-local
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
      const int gm = 16 * t + c;
      real vin[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) vin[r] = conv_act(acc[t][r] + bv[t], act);
      const real o = fmax(fmax(vin[0], vin[1]), fmax(vin[2], vin[3]));
      int oi = 4;
      if constexpr (IDX) {
#pragma unroll
        for (int r = 3; r >= 0; --r) oi = pool_approx(o, vin[r]) ? r : oi;
      }
      const int64_t win = 4 * sl + q;
      if (gm < Mp && win < nwin) {
        Out[gm + (int64_t)Mp * win] = o;
        if constexpr (IDX) Idx[gm + (int64_t)Mp * win] = (uint8_t)oi;
      }
    }
    if (more) {
#pragma unroll
      for (int s = 0; s < NS; ++s) fcur[s] = fnext[s];
    }
  }
}

static bool conv_first_applies(const ConvGeom& g, int COUTp) {
  // (tap offsets are packed into a byte each, biased by 128; up to 9 k steps = 36 taps: two slices of patch fragments live in
  // registers, 13 k steps spilled at the three-waves-per-SIMD budget)
  return g.Kvalid <= 36 && COUTp <= 64 && g.sden_w == 1 && g.sden_h == 1 && g.Cp <= 64 && g.pad_w < 100 && g.pad_h < 100 &&
         (g.KW - 1) * g.dil_w < 100 && (g.KH - 1) * g.dil_h < 100;
}

template <int NS, bool IDX>
static void launch_conv_first_ns(hipStream_t st, const real* Wp, const real* bp, const real* In, real* Out, uint8_t* Idx,
                                 const ConvGeom& g, int COUTp, int Kp, int64_t npos, int act) {
  const int ntm = (COUTp + 15) / 16;
  const int64_t nslice = (npos + 15) / 16;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (nslice + 3) / 4));
#define SI_FIRST_CASE(NTM)                                                                                                  \
  {                                                                                                                         \
    constexpr size_t lds = (size_t)4 * NS * (NTM * 16 + 16) * sizeof(real);                                               \
    hipLaunchKernelGGL((conv_first_pool_kernel<NS, NTM, IDX>), dim3(grid), dim3(256), lds, st, Wp, COUTp, In, Out, bp, g,   \
                       npos, Kp, act, Idx);                                                                                 \
  }
  switch (ntm) {
    case 1: SI_FIRST_CASE(1) break;
    case 2: SI_FIRST_CASE(2) break;
    case 3: SI_FIRST_CASE(3) break;
    default: SI_FIRST_CASE(4) break;
  }
#undef SI_FIRST_CASE
}
template <bool IDX>
static void launch_conv_first(hipStream_t st, const real* Wp, const real* bp, const real* In, real* Out, uint8_t* Idx,
                              const ConvGeom& g, int COUTp, int Kp, int64_t npos, int act) {
  const int ns = (g.Kvalid + 3) / 4;   // k steps that carry taps; the instantiation rounds up, the surplus reads zeros
  if (ns <= 5)
    launch_conv_first_ns<5, IDX>(st, Wp, bp, In, Out, Idx, g, COUTp, Kp, npos, act);
  else
    launch_conv_first_ns<9, IDX>(st, Wp, bp, In, Out, Idx, g, COUTp, Kp, npos, act);
}

#ifndef SI_CONV_F32   // (reverse sweep: fp64 only)
// part[split][m + Mp*k'] = sum over the split's positions of Delta[m + Mp*pos] * patch[k', pos]
// NOEDGE: npos % 16 == 0, so every split is a whole number of k tiles (the ragged-tile code is not compiled in)
template <int BM, int BN, int WM, int WN, int MINW, bool NOEDGE>
__global__ __launch_bounds__(64 * WM * WN, MINW) void conv_dw_kernel(const real* __restrict__ Delta, int Mp,
                                                                     const real* __restrict__ T, real* __restrict__ part,
                                                                     ConvGeom g, int64_t npos, int Kp, int64_t ksplit, int nMt,
                                                                     int nNt) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  using SA = Stager<BM, 0, NT, true>;
  using SB = GatherN<BN, NT>;
  extern __shared__ real smem[];
  const int mt = (int)(blockIdx.x % nMt), nt = (int)(blockIdx.x / nMt);
  const int64_t split = blockIdx.y;
  const int64_t k0 = split * ksplit;
  int64_t klen = npos - k0;
  if (klen > ksplit) klen = ksplit;
  if (klen <= 0 || nt >= nNt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int m0 = mt * BM, n0 = nt * BN;
  r4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (r4){0.0, 0.0, 0.0, 0.0};
  SA sa;
  SB sb;
  sa.init(Delta, Mp, m0, Mp, k0, tid);
  sb.init(T, g, n0, k0, k0 + klen, tid);
  gemm_mainloop<BM, BN, WM, WN, NOEDGE>(sa, sb, smem, (int)((klen + 15) / 16), klen, wm, wn, lane, acc, 0);
  gemm_epilogue<BM, BN, WM, WN, true, 2 * (SA::LDS_ELEMS + SB::LDS_ELEMS)>(
      acc, smem, part + split * (int64_t)Mp * Kp, (int64_t)Mp, m0, (int64_t)n0, Mp, (int64_t)Kp, wm, wn, lane, wave,
      [&](real v, int64_t, int) { return v; });
}


#endif
template <int BM, bool CELLU, bool DEN, bool BIASACT>
static void launch_conv_gemm_bm(hipStream_t st, const real* Wp, int Mp, const real* T, real* Out, const real* bias,
                                const ConvGeom& g, int64_t npos, int Kp, int act) {
  constexpr int BN = 128, WM = 2, WN = 4, NT = 512;
  using SA = Stager<BM, 0, NT, true>;
  using SB = GatherK<BN, NT, CELLU, DEN>;
  constexpr size_t lds = 2 * (SA::LDS_ELEMS + SB::LDS_ELEMS) * sizeof(real);
  const int nMt = (Mp + BM - 1) / BM;
  const int64_t nNt = (npos + BN - 1) / BN;
  const int64_t grid = nNt >= 8 ? (nNt + 7) / 8 * nMt * 8 : nNt * nMt;
  auto kern = conv_gemm_kernel<BM, BN, WM, WN, 4, CELLU, DEN, BIASACT>;
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, Wp, Mp, T, Out, bias, g, npos, Kp, act, nMt, nNt);
}

template <bool DEN, bool BIASACT>
static void launch_conv_gemm(hipStream_t st, const real* Wp, int Mp, const real* T, real* Out, const real* bias,
                             const ConvGeom& g, int64_t npos, int Kp, int act) {
  const bool cellu = g.Cp % 16 == 0;
  const int bm = conv_pick_bm(Mp);
#define SI_CONV_CASE(BM)                                                                                     \
  if (cellu)                                                                                                 \
    launch_conv_gemm_bm<BM, true, DEN, BIASACT>(st, Wp, Mp, T, Out, bias, g, npos, Kp, act);                \
  else                                                                                                       \
    launch_conv_gemm_bm<BM, false, DEN, BIASACT>(st, Wp, Mp, T, Out, bias, g, npos, Kp, act)
  if (bm == 64) {
    SI_CONV_CASE(64);
  } else if (bm == 96) {
    SI_CONV_CASE(96);
  } else {
    SI_CONV_CASE(128);
  }
#undef SI_CONV_CASE
}

// H[e] = act(H[e]) for the activations the GEMM epilogues do not carry (kernels_gemm.h)
__global__ __launch_bounds__(256) void act_inplace_kernel(real* __restrict__ H, int64_t n, int act) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) H[e] = (real)act_full((double)H[e], act);
}
void launch_act_inplace(hipStream_t st, real* H, int64_t n, int act) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  hipLaunchKernelGGL(act_inplace_kernel, dim3((unsigned)(b < 1 ? 1 : b)), dim3(256), 0, st, H, n, act);
}

void launch_conv_forward(hipStream_t st, const real* Wp, const real* bp, const real* In, real* Out, const ConvGeom& g,
                         int COUTp, int Kp, int64_t npos, int act) {
  if (act_is_extra(act)) {   // bias in the GEMM epilogue, the later activation elementwise (pad channels: act(0), as in the fused case)
    launch_conv_gemm<false, true>(st, Wp, COUTp, In, Out, bp, g, npos, Kp, SI_ACT_IDENTITY);
    launch_act_inplace(st, Out, (int64_t)COUTp * npos, act);
    return;
  }
  launch_conv_gemm<false, true>(st, Wp, COUTp, In, Out, bp, g, npos, Kp, act);
}

template <int BM, bool CELLU, bool IDX>
static void launch_conv_pool_bm(hipStream_t st, const real* Wp, int Mp, const real* T, real* Out, const real* bias,
                                const ConvGeom& g, int64_t npos, int Kp, int act, uint8_t* Idx) {
  constexpr int BN = 128, WM = 2, WN = 4, NT = 512;
  using SA = Stager<BM, 0, NT, true>;
  using SB = GatherK<BN, NT, CELLU, false, true>;
  constexpr size_t lds = 2 * (SA::LDS_ELEMS + SB::LDS_ELEMS) * sizeof(real);
  const int nMt = (Mp + BM - 1) / BM;
  const int64_t nNt = (npos + BN - 1) / BN;
  const int64_t grid = nNt >= 8 ? (nNt + 7) / 8 * nMt * 8 : nNt * nMt;
  auto kern = conv_gemm_pool_kernel<BM, BN, WM, WN, 4, CELLU, IDX>;
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, Wp, Mp, T, Out, bias, g, npos, Kp, act, nMt, nNt, Idx);
}
template <bool IDX>
static void launch_conv_pool_any(hipStream_t st, const real* Wp, const real* bp, const real* In, real* Out, const ConvGeom& g,
                                 int COUTp, int Kp, int64_t npos, int act, uint8_t* Idx) {
  const bool cellu = g.Cp % 16 == 0;
  const int bm = conv_pick_bm(COUTp);
#define SI_POOL_CASE(BM)                                                                         \
  if (cellu)                                                                                     \
    launch_conv_pool_bm<BM, true, IDX>(st, Wp, COUTp, In, Out, bp, g, npos, Kp, act, Idx);      \
  else                                                                                           \
    launch_conv_pool_bm<BM, false, IDX>(st, Wp, COUTp, In, Out, bp, g, npos, Kp, act, Idx)
  if (bm == 64) {
    SI_POOL_CASE(64);
  } else if (bm == 96) {
    SI_POOL_CASE(96);
  } else {
    SI_POOL_CASE(128);
  }
#undef SI_POOL_CASE
}
// conv + bias + act + MaxPool((2, 2), stride 2) -> pooled CWHN tensor; needs even Wo and Ho (the caller checks)
void launch_conv_forward_pool2(hipStream_t st, const real* Wp, const real* bp, const real* In, real* Out, const ConvGeom& g,
                               int COUTp, int Kp, int64_t npos, int act) {
  if (act_is_extra(act)) {   // increasing activations commute with the maximum: pool the pre-activations, finish on the pooled tensor
    launch_conv_forward_pool2(st, Wp, bp, In, Out, g, COUTp, Kp, npos, SI_ACT_IDENTITY);
    launch_act_inplace(st, Out, (int64_t)COUTp * (npos / 4), act);
    return;
  }
  if (conv_first_applies(g, COUTp)) {
    launch_conv_first<false>(st, Wp, bp, In, Out, nullptr, g, COUTp, Kp, npos, act);
    return;
  }
  launch_conv_pool_any<false>(st, Wp, bp, In, Out, g, COUTp, Kp, npos, act, nullptr);
}
// the same in GRADIENT mode (identity / relu / tanh / sigmoid only): also the window index the reverse sweep needs (one byte per
// pooled element, Idx[m + COUTp * window]); the un-pooled activation is neither stored nor read again
void launch_conv_forward_pool2_idx(hipStream_t st, const real* Wp, const real* bp, const real* In, real* Out, uint8_t* Idx,
                                   const ConvGeom& g, int COUTp, int Kp, int64_t npos, int act) {
  if (conv_first_applies(g, COUTp)) {
    launch_conv_first<true>(st, Wp, bp, In, Out, Idx, g, COUTp, Kp, npos, act);
    return;
  }
  launch_conv_pool_any<true>(st, Wp, bp, In, Out, g, COUTp, Kp, npos, act, Idx);
}

#ifndef SI_CONV_F32   // (reverse sweep: fp64 only)
void launch_conv_backward_data(hipStream_t st, const real* Wt, const real* Delta, real* dX, const ConvGeom& gT, int CINp,
                               int KpT, int64_t npos_in) {
  if (gT.sden_w > 1 || gT.sden_h > 1)
    launch_conv_gemm<true, false>(st, Wt, CINp, Delta, dX, nullptr, gT, npos_in, KpT, 0);
  else
    launch_conv_gemm<false, false>(st, Wt, CINp, Delta, dX, nullptr, gT, npos_in, KpT, 0);
}

// A FIRST conv layer (few input channels: Kp <= 64 taps, COUTp <= 64 rows) gives the weight-gradient GEMM ONE output tile
// and a k range of millions of positions: it is a stream over Delta, bound by the latency of one workgroup's two-deep
// pipeline (1.19 ms at 1.9 TB/s at the cfg4 CNN with two 128-column workgroups per CU).  Such a layer takes 64-column
// tiles (41 KB of LDS, half the idle MFMA columns) at THREE workgroups per CU.
static bool conv_dw_narrow(int COUTp, int Kp) { return Kp <= 64 && COUTp <= 64; }

int conv_dw_splits(int COUTp, int Kp, int64_t npos, int num_cu, int64_t* ksplit_out) {
  const int bm = conv_pick_bm(COUTp);
  const bool narrow = conv_dw_narrow(COUTp, Kp);
  const int64_t tiles = (int64_t)((COUTp + bm - 1) / bm) * (narrow ? 1 : (Kp + 127) / 128);
  // two (narrow: three) workgroups per CU are resident: round DOWN so that tiles * splits fits one residency round (rounding
  // up left 3..28 workgroups for a second round that ran almost alone -- 515 / 522 / 540 workgroups on 512 slots at cfg4)
  int64_t ns = ((int64_t)num_cu * (narrow ? 3 : 2)) / tiles;
  const int64_t maxsplit = (npos + 255) / 256;
  ns = std::max<int64_t>(1, std::min(ns, maxsplit));
  const int64_t ks = ((npos + ns - 1) / ns + 15) / 16 * 16;
  *ksplit_out = ks;
  return (int)((npos + ks - 1) / ks);
}

// upper bound of conv_dw_splits over every position count up to npos (scratch sizing: the split count is not monotonic in
// npos, see backward_weight_part_elems)
int conv_dw_max_splits(int COUTp, int Kp, int64_t npos, int num_cu) {
  const int bm = conv_pick_bm(COUTp);
  const bool narrow = conv_dw_narrow(COUTp, Kp);
  const int64_t tiles = (int64_t)((COUTp + bm - 1) / bm) * (narrow ? 1 : (Kp + 127) / 128);
  const int64_t ns = ((int64_t)num_cu * (narrow ? 3 : 2)) / tiles;
  return (int)std::max<int64_t>(1, std::min(ns, (npos + 255) / 256));
}

template <int BM, int BN, int MINW>
static void launch_conv_dw_bm(hipStream_t st, const real* Delta, int Mp, const real* In, real* part, const ConvGeom& g,
                              int64_t npos, int Kp, int nsplit, int64_t ksplit) {
  constexpr int WM = 2, WN = 4, NT = 512;
  using SA = Stager<BM, 0, NT, true>;
  using SB = GatherN<BN, NT>;
  constexpr size_t lds = 2 * (SA::LDS_ELEMS + SB::LDS_ELEMS) * sizeof(real);
  const int nMt = (Mp + BM - 1) / BM, nNt = (Kp + BN - 1) / BN;
  if (npos % 16 == 0) {   // (ksplit is a multiple of 16 by construction)
    auto kern = conv_dw_kernel<BM, BN, WM, WN, MINW, true>;
    static LdsOptIn optin;
    optin.ensure(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nMt * nNt), (unsigned)nsplit), dim3(NT), lds, st, Delta, Mp, In, part, g, npos, Kp,
                       ksplit, nMt, nNt);
  } else {
    auto kern = conv_dw_kernel<BM, BN, WM, WN, MINW, false>;
    static LdsOptIn optin;
    optin.ensure(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nMt * nNt), (unsigned)nsplit), dim3(NT), lds, st, Delta, Mp, In, part, g, npos, Kp,
                       ksplit, nMt, nNt);
  }
}

void launch_conv_backward_weight(hipStream_t st, const real* Delta, const real* In, real* part, const ConvGeom& g,
                                 int COUTp, int Kp, int64_t npos, int nsplit, int64_t ksplit) {
  if (conv_dw_narrow(COUTp, Kp)) {
    launch_conv_dw_bm<64, 64, 6>(st, Delta, COUTp, In, part, g, npos, Kp, nsplit, ksplit);
    return;
  }
  switch (conv_pick_bm(COUTp)) {
    case 64: launch_conv_dw_bm<64, 128, 4>(st, Delta, COUTp, In, part, g, npos, Kp, nsplit, ksplit); break;
    case 96: launch_conv_dw_bm<96, 128, 4>(st, Delta, COUTp, In, part, g, npos, Kp, nsplit, ksplit); break;
    default: launch_conv_dw_bm<128, 128, 4>(st, Delta, COUTp, In, part, g, npos, Kp, nsplit, ksplit); break;
  }
}

// ------------------------------------------------------------------------------------------------ index kernels

#endif
// Wp[co + COUTp*k'] = w[(KW-1-a) + KW*((KH-1-c) + KH*(cin + CIN*co))],  k' = cin + CINp*(a + KW*c); zeros elsewhere.
// bp[co] = b[co] (0 for the pad channel).
__global__ __launch_bounds__(256) void conv_pack_kernel(const real* __restrict__ w, const real* __restrict__ b,
                                                        real* __restrict__ Wp, real* __restrict__ bp, int KW, int KH, int CIN,
                                                        int COUT, int CINp, int COUTp, int Kp) {
  const int64_t total = (int64_t)COUTp * Kp;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int co = (int)(e % COUTp), kp = (int)(e / COUTp);
    const int cell = kp / CINp, cin = kp - cell * CINp;
    const int c = cell / KW, a = cell - c * KW;
    real v = 0.0;
    if (co < COUT && cin < CIN && c < KH) v = w[(KW - 1 - a) + KW * ((KH - 1 - c) + KH * (cin + (int64_t)CIN * co))];
    Wp[e] = v;
    if (e < COUTp) bp[e] = e < COUT ? b[e] : 0.0;
  }
}
void launch_conv_pack(hipStream_t st, const real* w, const real* b, real* Wp, real* bp, int KW, int KH, int CIN, int COUT,
                      int CINp, int COUTp, int Kp) {
  hipLaunchKernelGGL(conv_pack_kernel, dim3(idx_grid((int64_t)COUTp * Kp)), dim3(256), 0, st, w, b, Wp, bp, KW, KH, CIN, COUT,
                     CINp, COUTp, Kp);
}

#ifndef SI_CONV_F32   // (reverse sweep: fp64 only)
// data gradient: dX[cin, pin] = sum_{co, a', c'} Wt[cin + CINp*k''] * Delta[co, (pin + pad' ... )],  k'' = co + COUTp*(a' + KW*c'),
// Wt = w[a' + KW*(c' + KH*(cin + CIN*co))]  (the forward kernel flipped twice = not flipped)
__global__ __launch_bounds__(256) void conv_pack_t_kernel(const real* __restrict__ w, real* __restrict__ Wt, int KW, int KH,
                                                          int CIN, int COUT, int CINp, int COUTp, int KpT) {
  const int64_t total = (int64_t)CINp * KpT;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int cin = (int)(e % CINp), kp = (int)(e / CINp);
    const int cell = kp / COUTp, co = kp - cell * COUTp;
    const int c = cell / KW, a = cell - c * KW;
    real v = 0.0;
    if (co < COUT && cin < CIN && c < KH) v = w[a + KW * (c + KH * (cin + (int64_t)CIN * co))];
    Wt[e] = v;
  }
}
void launch_conv_pack_t(hipStream_t st, const real* w, real* Wt, int KW, int KH, int CIN, int COUT, int CINp, int COUTp,
                        int KpT) {
  hipLaunchKernelGGL(conv_pack_t_kernel, dim3(idx_grid((int64_t)CINp * KpT)), dim3(256), 0, st, w, Wt, KW, KH, CIN, COUT, CINp,
                     COUTp, KpT);
}

// gw[(KW-1-a) + KW*((KH-1-c) + KH*(cin + CIN*co))] = sum_split part[split][co + COUTp*k'],  fixed order
__global__ __launch_bounds__(256) void conv_unpack_dw_kernel(const real* __restrict__ part, int nsplit, real* __restrict__ gw,
                                                             int KW, int KH, int CIN, int COUT, int CINp, int COUTp, int Kp) {
  // walks the SOURCE order (co fastest: coalesced reads of the nsplit partial planes, which are 14-768x the bytes of the
  // result) and scatters the 8-byte results; in destination order every read touched its own cache line (0.6 TB/s).
  // Block = 32 source elements x 8 split phases: phase j adds splits j, j + 8, ... in order, the eight phase sums are added
  // in a fixed tree (a first layer has 3072 elements and 768 splits: one thread per element was 768 dependent loads).
  __shared__ real red[8][33];
  const int64_t plane = (int64_t)COUTp * Kp;
  const int il = threadIdx.x & 31, cl = threadIdx.x >> 5;
  for (int64_t s0 = (int64_t)blockIdx.x * 32; s0 < plane; s0 += (int64_t)gridDim.x * 32) {
    const int64_t src = s0 + il;
    real s = 0.0;
    if (src < plane)
      for (int sp = cl; sp < nsplit; sp += 8) s += part[(int64_t)sp * plane + src];
    red[cl][il] = s;
    __syncthreads();
    if (cl == 0 && src < plane) {
      const real t = ((red[0][il] + red[1][il]) + (red[2][il] + red[3][il])) + ((red[4][il] + red[5][il]) + (red[6][il] + red[7][il]));
      const int co = (int)(src % COUTp), kp = (int)(src / COUTp);
      const int cell = kp / CINp, cin = kp - cell * CINp;
      const int c = cell / KW, a = cell - c * KW;
      if (co < COUT && cin < CIN && c < KH)   // (else: pad channel / pad taps)
        gw[(KW - 1 - a) + KW * ((KH - 1 - c) + KH * (cin + (int64_t)CIN * co))] = t;
    }
    __syncthreads();
  }
}
void launch_conv_unpack_dw(hipStream_t st, const real* part, int nsplit, real* gw, int KW, int KH, int CIN, int COUT,
                           int CINp, int COUTp, int Kp) {
  const int64_t blocks = std::min<int64_t>(8192, ((int64_t)COUTp * Kp + 31) / 32);
  hipLaunchKernelGGL(conv_unpack_dw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, nsplit, gw, KW, KH, CIN, COUT, CINp,
                     COUTp, Kp);
}

#endif
// (W, H, C, N) column-major  ->  channel-fastest with pitch Cp (pad channels zero)
__global__ __launch_bounds__(256) void whcn_to_cwhn_kernel(const real* __restrict__ X, real* __restrict__ Xc, int W, int H,
                                                           int C, int Cp, int64_t B) {
  const int64_t total = (int64_t)Cp * W * H * B;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int wh = W * H;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % Cp);
    const int64_t t = e / Cp;
    const int sp = (int)(t % wh);
    const int64_t n = t / wh;
    Xc[e] = c < C ? X[sp + (int64_t)wh * (c + (int64_t)C * n)] : 0.0;
  }
}
void launch_whcn_to_cwhn(hipStream_t st, const real* X, real* Xc, int W, int H, int C, int Cp, int64_t B) {
  hipLaunchKernelGGL(whcn_to_cwhn_kernel, dim3(idx_grid((int64_t)Cp * W * H * B)), dim3(256), 0, st, X, Xc, W, H, C, Cp, B);
}
// channel-fastest (pitch Cp) -> the reference's flatten order  w + W*(h + H*c)  per observation
__global__ __launch_bounds__(256) void cwhn_to_whcn_kernel(const real* __restrict__ Xc, real* __restrict__ X, int W, int H,
                                                           int C, int Cp, int64_t B) {
  const int64_t total = (int64_t)C * W * H * B;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int wh = W * H;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int sp = (int)(e % wh);
    const int64_t t = e / wh;
    const int c = (int)(t % C);
    const int64_t n = t / C;
    X[e] = Xc[c + (int64_t)Cp * (sp + (int64_t)wh * n)];
  }
}
void launch_cwhn_to_whcn(hipStream_t st, const real* Xc, real* X, int W, int H, int C, int Cp, int64_t B) {
  hipLaunchKernelGGL(cwhn_to_whcn_kernel, dim3(idx_grid((int64_t)C * W * H * B)), dim3(256), 0, st, Xc, X, W, H, C, Cp, B);
}

// MaxPool on the channel-fastest layout
__global__ __launch_bounds__(256) void maxpool_kernel(const real* __restrict__ In, real* __restrict__ Out, int Cp, int Wi, int Hi,
                                                      int Wo, int Ho, int PW, int PH, int sw, int sh, int64_t B) {
  const int64_t total = (int64_t)Cp * Wo * Ho * B;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % Cp);
    int64_t t = e / Cp;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int64_t n = t / Ho;
    const real* src = In + c + (int64_t)Cp * ((int64_t)Wi * Hi * n);
    real m = -__builtin_inf();
    for (int d = 0; d < PH; ++d)
      for (int a = 0; a < PW; ++a) {
        const real v = src[(int64_t)Cp * ((wo * sw + a) + Wi * (ho * sh + d))];
        m = v > m ? v : m;
      }
    Out[e] = m;
  }
}
void launch_maxpool(hipStream_t st, const real* In, real* Out, int Cp, int Wi, int Hi, int Wo, int Ho, int PW, int PH, int sw,
                    int sh, int64_t B) {
  hipLaunchKernelGGL(maxpool_kernel, dim3(idx_grid((int64_t)Cp * Wo * Ho * B)), dim3(256), 0, st, In, Out, Cp, Wi, Hi, Wo, Ho, PW,
                     PH, sw, sh, B);
}
#ifndef SI_CONV_F32   // (reverse sweep: fp64 only)
// MaxPool gradient [upstream NNlib 0.7.23 src/impl/pooling_direct.jl `∇maxpool_direct!`, from memory]: for each window the
// inputs are walked `for kh in 1:kernel_h, kw in 1:kernel_w` (kw fastest) and the window's gradient goes to the FIRST input
// with `y ≈ x` (`maxpool_already_chosen`; isapprox: rtol = sqrt(eps), atol = 0) -- ONE element per window, not every
// element equal to the maximum: exact ties (constant image regions -> conv output = bias) would otherwise count 4 times.
// does input (a, d) of the window whose first input is at `win0` (element stride `es`, row stride `rs`) receive the
// window's gradient?  x = that input's value, ymax = the window's stored maximum
__device__ __forceinline__ bool pool_chosen(const real* __restrict__ win0, int64_t es, int64_t rs, real x, real ymax, int a,
                                            int d, int PW) {
  if (!pool_approx(ymax, x)) return false;
  for (int dd = 0; dd <= d; ++dd) {
    const int alim = dd == d ? a : PW;
    for (int aa = 0; aa < alim; ++aa)
      if (pool_approx(ymax, win0[es * aa + rs * dd])) return false;   // an earlier input already took it
  }
  return true;
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const real* __restrict__ In, const real* __restrict__ Out,
                                                          const real* __restrict__ Gout, real* __restrict__ Gin, int Cp, int Wi,
                                                          int Hi, int Wo, int Ho, int PW, int PH, int sw, int sh, int64_t B) {
  const int64_t total = (int64_t)Cp * Wi * Hi * B;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int c = (int)(e % Cp);
    int64_t t = e / Cp;
    const int wi = (int)(t % Wi);
    t /= Wi;
    const int hi = (int)(t % Hi);
    const int64_t n = t / Hi;
    const real x = In[e];
    real gsum = 0.0;
    // windows (wo, ho) with wo*sw <= wi < wo*sw + PW
    const int wo_hi = wi / sw, ho_hi = hi / sh;
    for (int ho = ho_hi; ho >= 0 && ho * sh + PH > hi; --ho) {
      if (ho >= Ho) continue;
      for (int wo = wo_hi; wo >= 0 && wo * sw + PW > wi; --wo) {
        if (wo >= Wo) continue;
        const int64_t o = c + (int64_t)Cp * (wo + (int64_t)Wo * (ho + (int64_t)Ho * n));
        const real* win0 = In + c + (int64_t)Cp * ((wo * sw) + (int64_t)Wi * ((ho * sh) + (int64_t)Hi * n));
        if (pool_chosen(win0, Cp, (int64_t)Cp * Wi, x, Out[o], wi - wo * sw, hi - ho * sh, PW)) gsum += Gout[o];
      }
    }
    Gin[e] = gsum;
  }
}
void launch_maxpool_bwd(hipStream_t st, const real* In, const real* Out, const real* Gout, real* Gin, int Cp, int Wi, int Hi,
                        int Wo, int Ho, int PW, int PH, int sw, int sh, int64_t B) {
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(idx_grid((int64_t)Cp * Wi * Hi * B)), dim3(256), 0, st, In, Out, Gout, Gin, Cp, Wi,
                     Hi, Wo, Ho, PW, PH, sw, sh, B);
}

// D[e] = G[e] * act'(H[e])   (pre-activation gradient of a layer from the gradient of its output)
__global__ __launch_bounds__(256) void mul_dact_kernel(const real* __restrict__ G, const real* __restrict__ H, int64_t n, int act,
                                                       real* __restrict__ D) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) D[e] = G[e] * dact_full(H[e], act);
}
void launch_mul_dact(hipStream_t st, const real* G, const real* H, int64_t n, int act, real* D) {
  hipLaunchKernelGGL(mul_dact_kernel, dim3(idx_grid(n)), dim3(256), 0, st, G, H, n, act, D);
}

// D = G .* act'(H) and db[i] = sum_b D[i + rows*b] in ONE pass (the reverse sweep of a conv / dense layer needs both; as
// separate kernels the row sum re-read the whole Delta tensor -- 2.1 GB behind the first conv layer of the cfg4 CNN).
// POOL = true additionally folds the MaxPool gradient in front of it: G is then the gradient of the POOLED tensor and the
// kernel routes it to the window maxima itself (H is the pool's input = the conv layer's output, already being read for
// act'), which saves writing and re-reading the un-pooled gradient.  rows = channels (fastest index), columns = pixels.
// Fixed-order sums: 4 column phases per chunk, then the chunks in 16 strided groups, each in order (bit-reproducible).
template <bool POOL>
__global__ __launch_bounds__(256) void dact_rowsum_kernel(const real* __restrict__ G, const real* __restrict__ H,
                                                          const real* __restrict__ PoolOut, int rows, int64_t ncols, int64_t per,
                                                          int act, real* __restrict__ D, real* __restrict__ part, int Wi, int Hi,
                                                          int Wo, int Ho, int PW, int PH, int sw, int sh) {
  __shared__ real red[4][64];
  const int il = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il;
  const int64_t b0 = (int64_t)blockIdx.y * per;
  int64_t b1 = b0 + per;
  if (b1 > ncols) b1 = ncols;
  real s = 0.0;
  if (i < rows) {
    int64_t b = b0 + cl;
    int wi = 0, hi = 0, dw = 0, dh = 0;
    int64_t n = 0, dn = 0;
    if constexpr (POOL) {  // pixel (wi, hi, image n) of column b, advanced by 4 columns per iteration (one carry per digit)
      wi = (int)(b % Wi);
      const int64_t t = b / Wi;
      hi = (int)(t % Hi);
      n = t / Hi;
      dw = 4 % Wi;
      dh = (4 / Wi) % Hi;
      dn = 4 / (Wi * Hi);
    }
#pragma unroll 4
    for (; b < b1; b += 4) {
      const int64_t off = i + (int64_t)rows * b;
      const real x = H[off];
      real g;
      if constexpr (POOL) {
        g = 0.0;  // the first input (kw fastest) that is ≈ the window's maximum receives its gradient [upstream NNlib, pool_chosen]
        for (int ho = hi / sh; ho >= 0 && ho * sh + PH > hi; --ho) {
          if (ho >= Ho) continue;
          for (int wo = wi / sw; wo >= 0 && wo * sw + PW > wi; --wo) {
            if (wo >= Wo) continue;
            const int64_t o = i + (int64_t)rows * (wo + (int64_t)Wo * (ho + (int64_t)Ho * n));
            const real* win0 = H + i + (int64_t)rows * ((wo * sw) + (int64_t)Wi * ((ho * sh) + (int64_t)Hi * n));
            if (pool_chosen(win0, rows, (int64_t)rows * Wi, x, PoolOut[o], wi - wo * sw, hi - ho * sh, PW)) g += G[o];
          }
        }
        wi += dw;
        hi += dh;
        n += dn;
        if (wi >= Wi) {
          wi -= Wi;
          ++hi;
        }
        if (hi >= Hi) {
          hi -= Hi;
          ++n;
        }
      } else {
        g = G[off];
      }
      const real d = g * dact_full(x, act);
      D[off] = d;
      s += d;
    }
  }
  red[cl][il] = s;
  __syncthreads();
  if (cl == 0 && i < rows) part[(int64_t)blockIdx.y * rows + i] = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
}
__global__ __launch_bounds__(1024) void rowsum_chunks_final_kernel(const real* __restrict__ part, int rows, int nchunks, int nout,
                                                                   real* __restrict__ db) {
  __shared__ real red[16][64];
  const int il = threadIdx.x & 63, cl = threadIdx.x >> 6;  // 16 strided groups of chunks, each summed in order
  const int i = blockIdx.x * 64 + il;
  real s = 0.0;
  if (i < nout) {
#pragma unroll 8
    for (int ch = cl; ch < nchunks; ch += 16) s += part[(int64_t)ch * rows + i];
  }
  red[cl][il] = s;
  __syncthreads();
  if (cl == 0 && i < nout) {
    real t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][il];
    db[i] = t;
  }
}
// chunks: enough workgroups (>= ~4096 of 4 waves) to run the stream at HBM speed whatever the row count is
static int dact_rowsum_chunks(int rows, int64_t ncols) {
  const int rb = (rows + 63) / 64;
  int64_t ch = std::max<int64_t>(64, (4096 + rb - 1) / rb);
  ch = std::min<int64_t>(ch, std::max<int64_t>(1, ncols / 16));
  return (int)ch;
}
size_t dact_rowsum_ws_elems(int max_rows) { return (size_t)4096 * 64 + (size_t)128 * (size_t)max_rows + 64; }
// D may alias G (in place).  db receives the first `nout` row sums (nout <= rows: pad channels are dropped).
void launch_mul_dact_rowsum(hipStream_t st, const real* G, const real* H, int rows, int64_t ncols, int act, real* D, real* part,
                            int nout, real* db) {
  const int nch = dact_rowsum_chunks(rows, ncols);
  const int64_t per = (ncols + nch - 1) / nch;
  hipLaunchKernelGGL(dact_rowsum_kernel<false>, dim3((rows + 63) / 64, nch), dim3(256), 0, st, G, H, nullptr, rows, ncols, per, act, D,
                     part, 0, 0, 0, 0, 0, 0, 0, 0);
  hipLaunchKernelGGL(rowsum_chunks_final_kernel, dim3((nout + 63) / 64), dim3(1024), 0, st, part, rows, nch, nout, db);
}
// MaxPool gradient + act' + bias row sum of the conv layer in front of the pool, one pass (D must not alias anything).
void launch_maxpool_bwd_dact_rowsum(hipStream_t st, const real* In, const real* Out, const real* Gout, real* D, int Cp, int Wi,
                                    int Hi, int Wo, int Ho, int PW, int PH, int sw, int sh, int64_t B, int act, real* part, int nout,
                                    real* db) {
  const int64_t ncols = (int64_t)Wi * Hi * B;
  const int nch = dact_rowsum_chunks(Cp, ncols);
  const int64_t per = (ncols + nch - 1) / nch;
  hipLaunchKernelGGL(dact_rowsum_kernel<true>, dim3((Cp + 63) / 64, nch), dim3(256), 0, st, Gout, In, Out, Cp, ncols, per, act, D,
                     part, Wi, Hi, Wo, Ho, PW, PH, sw, sh);
  hipLaunchKernelGGL(rowsum_chunks_final_kernel, dim3((nout + 63) / 64), dim3(1024), 0, st, part, Cp, nch, nout, db);
}

// Reverse sweep through Conv -> MaxPool((2, 2)) whose forward ran fused in gradient mode (conv_gemm_pool_kernel<IDX>): the Delta
// tensor of the conv layer from the POOLED gradient, the POOLED output and the byte index alone --
//   D[m, position k of window w] = (k == Idx[m, w]) ? G[m, w] * act'(Hp[m, w]) : 0,    db[m] = sum_w G[m, w] * act'(Hp[m, w])
// (the chosen input equals the pooled output, so act' can be rebuilt from it -- EXCEPT in a window where an EARLIER input is
// within sqrt(eps) of the maximum without being it: NNlib's rule routes the gradient to that input, the reference then
// evaluates act' at ITS value, this kernel at the maximum.  |Hp - h_chosen| <= 1.5e-8 |Hp|, so the entry of D is off by at
// most |G| * |act''| * 1.5e-8 |Hp| (3e-8 |G| for tanh); it shows where tanh / sigmoid saturate and many inputs of a window
// sit within 1e-8 of 1: 1e-9 of the largest gradient entry in tools/guard_fuzz_cnn.py, seed 74 case 122.  The un-fused
// route (later activations, odd sizes, other windows) keeps the un-pooled activation and is exact.)
// Reads 2 * 8 + 1 bytes per pooled element and
// writes the 4 * 8 bytes of D: 3.2 GB instead of 5.3 GB behind the first layer of the cfg4 CNN, and the forward no longer
// writes (2.1 GB) and re-reads (MaxPool pass, 2.1 GB) the un-pooled activation.  Same block shape and fixed-order sums as
// dact_rowsum_kernel: 64 rows x 4 window phases, chunks of windows in grid.y.
__global__ __launch_bounds__(256) void pool2_bwd_idx_kernel(const real* __restrict__ G, const real* __restrict__ Hp,
                                                            const uint8_t* __restrict__ Idx, int rows, int64_t nwin, int64_t per,
                                                            int act, real* __restrict__ D, real* __restrict__ part, int W2,
                                                            int H2) {
  __shared__ real red[4][64];
  const int il = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + il;
  const int64_t b0 = (int64_t)blockIdx.y * per;
  int64_t b1 = b0 + per;
  if (b1 > nwin) b1 = nwin;
  real s = 0.0;
  if (i < rows) {
    int64_t b = b0 + cl;
    // window (w2, h2, image n) of index b, advanced by 4 windows per iteration (one carry per digit)
    int w2 = (int)(b % W2);
    const int64_t t = b / W2;
    int h2 = (int)(t % H2);
    int64_t n = t / H2;
    const int dw = 4 % W2, dh = (4 / W2) % H2;
    const int64_t dn = 4 / ((int64_t)W2 * H2);
    const int Wf = 2 * W2, Hf = 2 * H2;
#pragma unroll 2
    for (; b < b1; b += 4) {
      const int64_t off = i + (int64_t)rows * b;
      const real d = G[off] * conv_dact(Hp[off], act);
      const int k = Idx[off];
      s += k < 4 ? d : 0.0;   // (k = 4: no input was ≈ the maximum, e.g. a NaN -- nothing is routed, as in the reference)
      const int64_t p00 = (2 * w2) + (int64_t)Wf * ((2 * h2) + (int64_t)Hf * n);
      real* dst = D + i + (int64_t)rows * p00;
      dst[0] = k == 0 ? d : 0.0;
      dst[rows] = k == 1 ? d : 0.0;
      dst[(int64_t)rows * Wf] = k == 2 ? d : 0.0;
      dst[(int64_t)rows * (Wf + 1)] = k == 3 ? d : 0.0;
      w2 += dw;
      h2 += dh;
      n += dn;
      if (w2 >= W2) {
        w2 -= W2;
        ++h2;
      }
      if (h2 >= H2) {
        h2 -= H2;
        ++n;
      }
    }
  }
  red[cl][il] = s;
  __syncthreads();
  if (cl == 0 && i < rows) part[(int64_t)blockIdx.y * rows + i] = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
}
void launch_pool2_bwd_idx(hipStream_t st, const real* G, const real* Hp, const uint8_t* Idx, real* D, int Cp, int W2, int H2,
                          int64_t B, int act, real* part, int nout, real* db) {
  const int64_t nwin = (int64_t)W2 * H2 * B;
  const int nch = dact_rowsum_chunks(Cp, nwin);
  const int64_t per = (nwin + nch - 1) / nch;
  hipLaunchKernelGGL(pool2_bwd_idx_kernel, dim3((Cp + 63) / 64, nch), dim3(256), 0, st, G, Hp, Idx, Cp, nwin, per, act, D, part, W2, H2);
  hipLaunchKernelGGL(rowsum_chunks_final_kernel, dim3((nout + 63) / 64), dim3(1024), 0, st, part, Cp, nch, nout, db);
}

#endif
// Dense layer with a NARROW output (out <= 16: the 10-class head of a CNN) on a small batch.  The MFMA kernel gives such a
// layer one 32-row tile per 128 columns -- 32 workgroups at B = 4096, each walking all of `in` as one dependent chain of k
// tiles (0.22 ms for 4096 -> 10 at B = 4096, 0.6 TB/s).  Here a WAVE owns CW columns: lanes stride over k (coalesced reads
// of the activation column and of W's 8*out-byte rows), out * CW accumulators per lane, one butterfly reduction at the end.
// Fixed summation order (k = lane, lane + 64, ... then the xor butterfly): bit-reproducible.
template <int CW>
__global__ __launch_bounds__(256) void dense_narrow_kernel(const real* __restrict__ W, const real* __restrict__ bias,
                                                           const real* __restrict__ Hin, real* __restrict__ Hout, int out, int in,
                                                           int64_t B, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b0 = ((int64_t)blockIdx.x * 4 + wave) * CW;
  if (b0 >= B) return;
  real acc[CW][16];
#pragma unroll
  for (int c = 0; c < CW; ++c)
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[c][o] = 0.0;
  const real* hcol[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) hcol[c] = Hin + (int64_t)in * (b0 + c < B ? b0 + c : B - 1);
  for (int k = lane; k < in; k += 64) {
    real h[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) h[c] = hcol[c][k];
    const real* wk = W + (int64_t)out * k;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const real wv = o < out ? wk[o] : 0.0;
#pragma unroll
      for (int c = 0; c < CW; ++c) acc[c][o] = fma(wv, h[c], acc[c][o]);
    }
  }
#pragma unroll
  for (int c = 0; c < CW; ++c)
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      real v = acc[c][o];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      acc[c][o] = v;
    }
  if (lane < 16 * CW) {
    const int c = lane >> 4, o = lane & 15;
    real v = 0.0;
#pragma unroll
    for (int cc = 0; cc < CW; ++cc)
#pragma unroll
      for (int oo = 0; oo < 16; ++oo)
        if (cc == c && oo == o) v = acc[cc][oo];
    if (o < out && b0 + c < B) Hout[o + (int64_t)out * (b0 + c)] = (real)act_full((double)(v + bias[o]), act);
  }
}
// true when the narrow kernel is the better choice: few MFMA workgroups and a long k chain
#ifndef SI_CONV_F32
bool dense_narrow_applies(int out, int in, int64_t B, int num_cu) { return out <= 16 && in >= 256 && (B + 127) / 128 < 2 * (int64_t)num_cu; }
#endif
void launch_dense_narrow(hipStream_t st, const real* W, const real* bias, const real* Hin, real* Hout, int out, int in, int64_t B,
                         int act) {
  constexpr int CW = 2;
  const int64_t blocks = (B + 4 * CW - 1) / (4 * CW);
  hipLaunchKernelGGL(dense_narrow_kernel<CW>, dim3((unsigned)blocks), dim3(256), 0, st, W, bias, Hin, Hout, out, in, B, act);
}

}  // namespace si
