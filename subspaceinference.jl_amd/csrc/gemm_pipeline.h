// The fp64 MFMA GEMM pipeline shared by the reverse-sweep kernels (kernels_bwd.hip) and the implicit-GEMM convolution
// kernels (kernels_conv.hip): 8 waves, 16-deep k tiles, two LDS buffers, ONE barrier per tile, the MFMAs of a tile split
// around its LDS / global traffic (same schedule as the forward kernel, kernels_gemm.hip), v_mfma_f64_16x16x4_f64.
//
// An operand is described by a STAGER type: it owns the global -> register -> LDS path of one operand tile
// (R rows x 16 k values per k tile) and leaves one of two LDS images, both conflict-free for the ds_read_b64 operand reads:
//   LAY = 0 "row-fast"  X[r + ld*k]  -> LDS [k][R+16]          LAY = 1 "k-fast"  X[k + ld*r] -> LDS [r][18]
// Interface: static LAY, LDS_ELEMS, RP, KP;  load(kt), store(dst), load_edge(kt, klen), store_edge(dst, kt, klen).
// `Stager` below reads a plain matrix; kernels_conv.hip adds stagers that gather im2col patches on the fly.
//
// The element type is a macro (default double, namespace gp64): kernels_conv.hip is compiled a second time with
// -DSI_CONV_F32 (float operands on v_mfma_f32_16x16x4_f32 -- the same 16 x 16 x 4 shape and lane map, so every stager, LDS image
// and index map below serves both; namespace gp32) for compute_dtype = SI_F32 on Conv chains.
#pragma once
#include <type_traits>

#include "kernels_gemm.h"

#ifdef SI_CONV_F32
#define SI_GP_NS gp32
#define SI_GP_REAL float
#define SI_GP_REAL2 float2
#define SI_GP_MAKE2 make_float2
#define SI_GP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
// component r of the accumulator in lane (q, c) is row 4 q + r of the 16 x 16 tile (the fp64 instruction: q + 4 r); a 2 x 2 pooling
// window keeps its four inputs in ONE lane either way: position p of a 16-position slice = window / input as below
#define SI_GP_ROW(q, r) (4 * (q) + (r))
#define SI_GP_PWIN(p) (((p) >> 2) & 3)
#define SI_GP_PIN(p) ((p) & 3)
#else
#define SI_GP_NS gp64
#define SI_GP_REAL double
#define SI_GP_REAL2 double2
#define SI_GP_MAKE2 make_double2
#define SI_GP_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)
#define SI_GP_ROW(q, r) ((q) + 4 * (r))
#define SI_GP_PWIN(p) ((p) & 3)
#define SI_GP_PIN(p) (((p) >> 2) & 3)
#endif

namespace si {
namespace SI_GP_NS {

typedef SI_GP_REAL real;
typedef SI_GP_REAL2 real2;
typedef real r4 __attribute__((ext_vector_type(4)));

// one operand tile: R rows (feature / batch index) x 16 k values
template <int R, int LAY_, int NT, bool VEC>
struct Stager {
  static constexpr int LAY = LAY_;
  static constexpr int E = VEC ? 2 : 1;
  static constexpr int NREG = (16 * R + NT * E - 1) / (NT * E);
  static constexpr int RP = R + 16, KP = 18;
  static constexpr int LDS_ELEMS = LAY == 0 ? 16 * RP : R * KP;
  // EXACT: the tile is a whole number of thread sweeps and consecutive slots of a thread differ by a constant step (in k for
  // LAY 0, in rows for LAY 1).  Then every slot is live and lds / kk (and go for LAY 0) of slot r follow from slot 0 --
  // the 128 x 128 kernels run on a 128-register budget and spilled into their k loop with per-slot copies of all of these.
  static constexpr bool EXACT = (16 * R) % (NT * E) == 0 && (LAY == 0 ? (NT * E) % R == 0 : (NT * E) % 16 == 0);
  static constexpr int STEP = LAY == 0 ? (NT * E) / R : (NT * E) / 16;   // k step (LAY 0) / row step (LAY 1) between slots
  static constexpr int NS = EXACT ? 1 : NREG;                           // per-slot copies actually kept
  static constexpr int NGO = (EXACT && LAY == 0) ? 1 : NREG;
  int go_[NGO], lds_[NS], kk_[NS];
  bool live_[NS];
  int ldi;
  real reg[NREG][E];
  const real* base;
  int64_t ld;

  __device__ __forceinline__ int go(int r) const {
    if constexpr (EXACT && LAY == 0) return go_[0] + r * STEP * ldi;
    else return go_[r];
  }
  __device__ __forceinline__ int lds(int r) const {
    if constexpr (EXACT) return lds_[0] + r * STEP * (LAY == 0 ? RP : KP);
    else return lds_[r];
  }
  __device__ __forceinline__ int kk(int r) const {
    if constexpr (EXACT) return kk_[0] + (LAY == 0 ? r * STEP : 0);
    else return kk_[r];
  }
  __device__ __forceinline__ bool live(int r) const {
    if constexpr (EXACT) return true;
    else return live_[r];
  }

  __device__ __forceinline__ void init(const real* X, int64_t ld_, int64_t r0, int64_t rmax, int64_t k0, int tid) {
    ld = ld_;
    ldi = (int)ld_;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      const int idx = (tid + NT * r) * E;
      if constexpr (LAY == 0) {
        const int rr = idx % R, k = idx / R;
        int64_t g = r0 + rr;
        if (g > rmax - E) g = rmax - E;  // clamped rows only feed outputs that are never stored
        if (g < 0) g = 0;
        if (r < NS) {
          kk_[r < NS ? r : 0] = k;
          live_[r < NS ? r : 0] = k < 16;
          lds_[r < NS ? r : 0] = k * RP + rr;
        }
        if (r < NGO) go_[r < NGO ? r : 0] = (int)(g - r0) + (int)ld * (k < 16 ? k : 0);
      } else {
        const int k = idx & 15, rr = idx >> 4;
        int64_t g = r0 + (rr < R ? rr : 0);
        if (g > rmax - 1) g = rmax - 1;
        if (r < NS) {
          kk_[r < NS ? r : 0] = k;
          live_[r < NS ? r : 0] = rr < R;
          lds_[r < NS ? r : 0] = rr * KP + k;
        }
        go_[r] = (int)(g - r0) * (int)ld + k;
      }
    }
    base = LAY == 0 ? X + r0 + ld * k0 : X + ld * r0 + k0;
  }
  // Hot path (every k tile but a ragged last one): loop-invariant per-thread offsets from a block-uniform pointer, no
  // clamps, no selects -- one address add per load and a bare ds_write per store, like the forward kernel.  (With the
  // edge logic inline the 16-deep tile cost ~60 VALU instructions per wave next to its 24 MFMAs, and the split-K weight
  // gradient ran at 51 TFLOP/s.)
  __device__ __forceinline__ void load(int kt) {
    const real* p = base + (LAY == 0 ? ld * 16 : (int64_t)16) * kt;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      if constexpr (VEC) {
        const real2 v = *reinterpret_cast<const real2*>(p + go(r));
        reg[r][0] = v.x;
        reg[r][1] = v.y;
      } else {
        reg[r][0] = p[go(r)];
      }
    }
  }
  __device__ __forceinline__ void store(real* dst) const {
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      if (!live(r)) continue;
      if constexpr (VEC)
        *reinterpret_cast<real2*>(dst + lds(r)) = SI_GP_MAKE2(reg[r][0], reg[r][1]);
      else
        dst[lds(r)] = reg[r][0];
    }
  }
  // Ragged last tile (klen % 16 != 0): k indices past the end are clamped to a legal address and zero-filled in LDS
  // (they would add into valid outputs).  klen_total = k values of this block's split.
  __device__ __forceinline__ void load_edge(int kt, int64_t klen_total) {
    const real* p = base + (LAY == 0 ? ld * 16 : (int64_t)16) * kt;
    const int64_t kmax = klen_total - (LAY == 1 ? E : 1) - (int64_t)kt * 16;  // last legal k (pair start) in this tile
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      int o = go(r);
      // (LAY 0: a slot past the tile (k >= 16) is not live: it points at k = 0 of the tile already and is never stored --
      // moved like a live one it would read up to 21 k values BEFORE a first tile, i.e. outside the operand)
      if ((LAY == 1 || live(r)) && kk(r) > kmax) o -= (int)(LAY == 0 ? ld : 1) * (int)(kk(r) - (kmax > 0 ? kmax : 0));
      if constexpr (VEC) {
        const real2 v = *reinterpret_cast<const real2*>(p + o);
        reg[r][0] = v.x;
        reg[r][1] = v.y;
      } else {
        reg[r][0] = p[o];
      }
    }
  }
  __device__ __forceinline__ void store_edge(real* dst, int kt, int64_t klen_total) const {
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      if (!live(r)) continue;
      const bool ok = (int64_t)kt * 16 + kk(r) < klen_total;
      if constexpr (VEC)
        *reinterpret_cast<real2*>(dst + lds(r)) = SI_GP_MAKE2(ok ? reg[r][0] : (real)0, ok ? reg[r][1] : (real)0);
      else
        dst[lds(r)] = ok ? reg[r][0] : (real)0;
    }
  }
};

// acc[a][b] += sum over the nk k tiles the two stagers deliver (klen = k values of this block's range; only the last tile
// may be ragged).  smem: [2][SA::LDS_ELEMS] then [2][SB::LDS_ELEMS].  wm / wn: this wave's position in the WM x WN grid.
// NOEDGE: the caller guarantees klen % 16 == 0 (no ragged tile): the edge variants of the stagers are not even compiled in.
template <int BM, int BN, int WM, int WN, bool NOEDGE = false, class SA, class SB>
__device__ __forceinline__ void gemm_mainloop(SA& sa, SB& sb, real* smem, int nk, int64_t klen, int wm, int wn, int lane,
                                              r4 (&acc)[BM / WM / 16][BN / WN / 16], int dbg) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int ALAY = SA::LAY, BLAY = SB::LAY;
  real* sAbuf = smem;
  real* sBbuf = smem + 2 * SA::LDS_ELEMS;
  const int q = lane >> 4, c = lane & 15;
  real fa[2][TM], fb[2][TN];
  const int aw = wm * (BM / WM) + c, bw = wn * (BN / WN) + c;
  const real* pa0 = sAbuf + (ALAY == 0 ? q * SA::RP + aw : aw * SA::KP + q);
  const real* pb0 = sBbuf + (BLAY == 0 ? q * SB::RP + bw : bw * SB::KP + q);
  auto read_frags = [&](auto BUF, auto S, auto SET) {
    constexpr int buf = decltype(BUF)::value, s = decltype(S)::value, set = decltype(SET)::value;
#pragma unroll
    for (int a = 0; a < TM; ++a)
      fa[set][a] = pa0[buf * SA::LDS_ELEMS + (ALAY == 0 ? 4 * s * SA::RP + a * 16 : a * 16 * SA::KP + 4 * s)];
#pragma unroll
    for (int b = 0; b < TN; ++b)
      fb[set][b] = pb0[buf * SB::LDS_ELEMS + (BLAY == 0 ? 4 * s * SB::RP + b * 16 : b * 16 * SB::KP + 4 * s)];
  };
  auto mfma_half = [&](auto SET, auto HALF) {
    constexpr int set = decltype(SET)::value, half = decltype(HALF)::value;
    constexpr int lo = half == 0 ? 0 : (TM * TN) / 2, hi = half == 0 ? (TM * TN) / 2 : TM * TN;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = lo; t < hi; ++t) {
      const int a = t / TN, b = t % TN;
      acc[a][b] = SI_GP_MFMA(fb[set][b], fa[set][a], acc[a][b]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // only the last tile of a split can be ragged; the choice is block-uniform (a scalar branch)
  const int edge_kt = (!NOEDGE && (klen & 15)) ? nk - 1 : -1;
  auto load_tile = [&](int kt) {
    if (!NOEDGE && kt == edge_kt) {
      sa.load_edge(kt, klen);
      sb.load_edge(kt, klen);
    } else {
      sa.load(kt);
      sb.load(kt);
    }
  };
  auto store_tile = [&](auto BUF, int kt) {
    constexpr int buf = decltype(BUF)::value;
    if (!NOEDGE && kt == edge_kt) {
      sa.store_edge(sAbuf + buf * SA::LDS_ELEMS, kt, klen);
      sb.store_edge(sBbuf + buf * SB::LDS_ELEMS, kt, klen);
    } else {
      sa.store(sAbuf + buf * SA::LDS_ELEMS);
      sb.store(sBbuf + buf * SB::LDS_ELEMS);
    }
  };
  // same pipeline as the forward kernel (kernels_gemm.hip tile_body): [half the MFMAs][LDS / global traffic][other half]
  auto tile_body = [&](auto BUF, auto NBUF, int kt) {
    mfma_half(I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    read_frags(BUF, I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(I0{}, I1{});
    __builtin_amdgcn_sched_barrier(0);

    mfma_half(I1{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    read_frags(BUF, I2{}, I0{});
    if (kt + 1 < nk) store_tile(NBUF, kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);

    mfma_half(I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    read_frags(BUF, I3{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(I0{}, I1{});
    __builtin_amdgcn_sched_barrier(0);

    mfma_half(I1{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (kt + 1 < nk) read_frags(NBUF, I0{}, I0{});
    if (kt + 2 < nk && !(dbg & 2)) load_tile(kt + 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
  };

  load_tile(0);
  store_tile(I0{}, 0);
  if (nk > 1) load_tile(1);
  __syncthreads();
  read_frags(I0{}, I0{}, I0{});
  for (int kt = 0; kt < nk; kt += 2) {
    tile_body(I0{}, I1{}, kt);
    if (kt + 1 < nk) tile_body(I1{}, I0{}, kt + 1);
  }

}

// Store the accumulators of a block: C[m + ldc*n] = f(acc, offset, m).  MFMA output is D[n = q + 4r][m = c] per 16 x 16
// tile; with VEC (even Mrows / ldc, 16-B aligned C) the tile is transposed through the idle staging LDS so that every
// store instruction writes 16 B per lane over whole rows of C.  SMEM_ELEMS = doubles of staging LDS available.
// f(value, element offset, m) -> value is applied per element (bias + activation, act' factor, ...).
template <int BM, int BN, int WM, int WN, bool VEC, int SMEM_ELEMS, class F>
__device__ __forceinline__ void gemm_epilogue(r4 (&acc)[BM / WM / 16][BN / WN / 16], real* smem, real* __restrict__ Cout,
                                              int64_t ldc, int m0, int64_t n0, int Mrows, int64_t Ncols, int wm, int wn, int lane,
                                              int wave, F f) {
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int WI = BM / WM;
  constexpr bool WIDE = VEC && ((16 * (WI / 2)) % 64 == 0) && (WM * WN * 16 * WI <= SMEM_ELEMS);   // WI = 32 / 48 / 64 / 128
  const int q = lane >> 4, c = lane & 15;
  const int mw0 = m0 + wm * WI;
  const int64_t nw0 = n0 + wn * (BN / WN);
  if constexpr (WIDE) {
    constexpr int CH_ROW = WI / 2, NCH = 16 * CH_ROW / 64;
    real* reg = smem + wave * (16 * WI);
    __syncthreads();  // all waves are done with the staging buffers (the loop's last barrier precedes the last reads)
#pragma unroll
    for (int bt = 0; bt < TN; ++bt) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) reg[SI_GP_ROW(q, r) * WI + a * 16 + c] = acc[a][bt][r];
#pragma unroll
      for (int p = 0; p < NCH; ++p) {
        const int chunk = p * 64 + lane;
        const int row = chunk / CH_ROW, col2 = chunk % CH_ROW;
        real2 v = *reinterpret_cast<const real2*>(reg + 2 * chunk);
        const int gm = mw0 + 2 * col2;
        const int64_t gn = nw0 + bt * 16 + row;
        if (gm + 1 < Mrows && gn < Ncols) {
          const int64_t off = gm + ldc * gn;
          v.x = f(v.x, off, gm);
          v.y = f(v.y, off + 1, gm + 1);
          *reinterpret_cast<real2*>(Cout + off) = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int gm = mw0 + a * 16 + c;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gn = nw0 + b * 16 + SI_GP_ROW(q, r);
          if (gm < Mrows && gn < Ncols) {
            const int64_t off = gm + ldc * gn;
            Cout[off] = f(acc[a][b][r], off, gm);
          }
        }
      }
    }
  }
}

}  // namespace SI_GP_NS
using namespace SI_GP_NS;
}  // namespace si
