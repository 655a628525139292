// K5  Dense-chain forward in fp64 on the gfx950 matrix cores.
// Replaces reference src/space_inference.jl:92-94: `model_re(in_model, new_W)` + `new_model(in_data)`, i.e. per
// Flux Dense layer  H' = act.(W*H .+ b)  with W (out x in) taken IN PLACE from the flat weight vector (the static
// layer-offset table replaces the per-call Flux.destructure/re copies of src/libs.jl:55-57).
//
// All matrices are column-major (Julia):  W[i + out*k],  Hin[k + in*b],  Hout[i + out*b].
//
// Kernel: LDS-tiled GEMM on v_mfma_f64_16x16x4_f64 (64 cycles per instruction per SIMD, 2048 flop: measured
// tools/mfma_f64_probe.hip).  One 16x16 MFMA tile has its ROWS on the batch index b and its COLUMNS on the
// feature index i, so that a wave stores 128-B contiguous runs of Hout.
//   A operand (16x4): lane l holds Hin[k = 4s + (l>>4)][b = l&15]     <- sH[b][k]   (row stride 18 doubles)
//   B operand (4x16): lane l holds   W[i = l&15][k = 4s + (l>>4)]     <- sW[k][i]   (row stride BM+16 doubles)
//   C/D: lane l, reg r holds D[b = (l>>4) + 4r][i = l&15]              (f64 map; NOT the f32 one)
// LDS row strides are chosen so that every ds_read_b64 of a 32-lane half hits 32 distinct bank pairs:
//   sW: (BM+16)*2 dwords == 32 (mod 64)  -> the two k rows of a half land in different halves of the bank row
//   sH: 18*2 = 36 dwords per b; 36*b mod 64 runs over all multiples of 4 for b = 0..15, +2 for the second k
// Staging is global -> registers -> LDS (branch-free: clamped address + select, 16-B accesses when the operands
// are 16-B aligned), double-buffered, one barrier per 16-deep k tile, software-pipelined under the MFMAs.
// Roofline: MFMA f64 (78.6 TFLOP/s); algorithmic flops = 2*out*in*B per layer.
#include <type_traits>

#include "kernels_gemm.h"

namespace si {

#ifdef SI_GEMM_DEBUG_KNOB
__device__ int si_gemm_dbg = 0;
static size_t si_gemm_lds_floor = 0;
__device__ long long si_gemm_stamps[4 * 16384];
#define SI_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) si_gemm_stamps[4 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SI_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ double apply_act(double v, int act) {
  switch (act) {
    case SI_ACT_RELU: return v > 0.0 ? v : 0.0;   // Flux relu(x) = max(0, x)
    case SI_ACT_TANH: return tanh(v);
    case SI_ACT_SIGMOID: return 1.0 / (1.0 + exp(-v));
    default: return v;
  }
}

// FUSE: the layer's output is not stored.  The NEXT (last) layer of the chain, out_last <= 4 wide, is applied in the
// epilogue: every wave reduces  sum_i Wlast[o][i] * act(H[i][b])  over its own features and writes one partial per
// (feature slot, o, b) to `part`; tail_sse_kernel adds the slots in fixed order.  This removes the out x B store, the
// whole GEMV-shaped last layer and its out x B re-read (cfg2: 768 MB written + 768 MB read per sample).
template <int BM, int BN, int WM, int WN, int MINW, bool VEC, bool KEDGE, bool FUSE>
__global__ __launch_bounds__(64 * WM * WN, MINW) void dense_f64_kernel(
    const double* __restrict__ W, const double* __restrict__ bias, const double* __restrict__ Hin,
    double* __restrict__ Hout, int out, int in, int64_t B, int act, int nMt, int64_t nNt,
    const double* __restrict__ Wlast, int out_last, double* __restrict__ part, ChainBatch cb) {
  // chain batching: blockIdx.y is the chain slot; every operand that differs per chain moves by its slot stride
  // (block-uniform scalar arithmetic; for a single chain all strides are unused)
  if (blockIdx.y != 0) {
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
#ifdef SI_GEMM_DEBUG_KNOB
  const int dbg = si_gemm_dbg;  // harness only, bit mask: 1 = every block loads tile (0,0) (L2-hot), 2 = no global loads in the k loop,
                                // 4 = no barrier in the k loop, 8 = no LDS stores in the k loop (4 and 8 give wrong results: timing only)
#else
  constexpr int dbg = 0;
#endif
  constexpr int NT = 64 * WM * WN;
  constexpr int BK = 16;
  constexpr int BMP = BM + 16;
  constexpr int BKP = 18;
  constexpr int TM = BM / WM / 16;  // feature tiles per wave
  constexpr int TN = BN / WN / 16;  // batch tiles per wave
  constexpr int E = VEC ? 2 : 1;    // doubles per staging access
  constexpr int WREGS = (BK * BM + NT * E - 1) / (NT * E);
  constexpr int HREGS = (BK * BN + NT * E - 1) / (NT * E);
  constexpr bool WRAG = (BK * BM) % (NT * E) != 0, HRAG = (BK * BN) % (NT * E) != 0;
  static_assert(BM % 32 == 0 && (BM / WM) % 16 == 0 && (BN / WN) % 16 == 0, "tile shape");
  extern __shared__ double smem[];
  double* sW = smem;                  // [2][BK][BMP]
  double* sH = smem + 2 * BK * BMP;   // [2][BN][BKP]

  // XCD-aware block -> tile map: blocks b and b+8 share an XCD (its L2); the nMt feature tiles of one batch panel go to
  // consecutive blocks of ONE XCD so that they find the panel's k tiles in that L2 while they walk it together.  That is a
  // locality HINT, not a guarantee: 64 resident workgroups per XCD span ~6.4 panels, and W (7.4 MB at 960 x 960) is
  // re-streamed by every round of workgroups -- the PMC passes count 2.76 GB per launch against 0.79 GB algorithmic (3.5x;
  // profiles/*_pmc_dense_main.json), served at ~1 TB/s mostly out of the Infinity Cache while the kernel is MFMA-bound.
  const int64_t bid = blockIdx.x;
  const int xcd = (int)(bid & 7);
  const int64_t j = bid >> 3;
  const int mt = (int)(j % nMt);
  const int64_t nt = (j / nMt) * 8 + xcd;
  if (nt >= nNt) return;  // uniform per block: whole workgroup exits before any barrier
  SI_STAMP(0);
  if ((dbg & 32) && bid < 512) {  // harness: stagger the first round of workgroups over ~one block time
    const int n = (int)((bid * 2654435761u) >> 24) & 31;
    const int unit = (dbg >> 8) ? (dbg >> 8) : 127;  // sleep quantum in 64-cycle units (dbg bits 8..)
    for (int i = 0; i < n * unit; ++i) __builtin_amdgcn_s_sleep(1);
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int i0 = mt * BM;
  const int64_t b0 = nt * BN;
  const int q = lane >> 4, c = lane & 15;

  d4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

  double wreg[WREGS][E], hreg[HREGS][E];
  const int nk = (in + BK - 1) / BK;

  // Per-thread staging coordinates, fixed for the whole k loop: a 32-bit element offset from a block-uniform base
  // pointer that advances by one k tile per iteration (so a load is one instruction, SGPR base + VGPR offset), and an
  // LDS element offset.  Feature / batch indices past the edge are CLAMPED to a valid row: such rows only feed output
  // elements that are never stored, so they need no zeroing.  Only a ragged k edge (in % 16 != 0, KEDGE) must be
  // zeroed, because it would add into valid outputs.
  int w_go[WREGS], w_lds[WREGS], w_k[WREGS];
#pragma unroll
  for (int r = 0; r < WREGS; ++r) {
    const int idx = (tid + NT * r) * E;
    const int ii = idx % BM, k = idx / BM;
    int gi = ((dbg & 1) ? 0 : i0) + ii;
    if (gi > out - E) gi = out - E;
    w_k[r] = k;
    w_go[r] = gi + out * (k < BK ? k : 0);
    w_lds[r] = k * BMP + ii;
  }
  int h_go[HREGS], h_lds[HREGS], h_kk[HREGS], h_b[HREGS];
  const int64_t bb = (dbg & 1) ? 0 : b0;
#pragma unroll
  for (int r = 0; r < HREGS; ++r) {
    const int idx = (tid + NT * r) * E;
    const int kk = idx & 15, b = idx >> 4;
    int64_t gb = bb + (b < BN ? b : 0);
    if (gb > B - 1) gb = B - 1;
    h_kk[r] = kk;
    h_b[r] = b;
    h_go[r] = (int)(gb - bb) * in + kk;
    h_lds[r] = b * BKP + kk;
  }
  const double* Wt = W;                         // + out*BK per k tile
  const double* Ht = Hin + (int64_t)in * bb;    // + BK per k tile

  auto load_tiles = [&](int kt) {
    const double* wp = Wt + (int64_t)out * BK * kt;
    const double* hp = Ht + BK * kt;
#pragma unroll
    for (int r = 0; r < WREGS; ++r) {
      int o = w_go[r];
      if constexpr (KEDGE) {  // clamp the k index of a ragged last tile to a legal column
        // (slots past the tile -- WRAG, k >= BK -- already point at k = 0 and are never stored: moving them as well
        // would put them up to (BK + 5 - in) columns BEFORE W, outside the weight vector for a first layer)
        const int kmax = in - 1 - kt * BK;
        if (w_k[r] > kmax && w_k[r] < BK) o -= out * (w_k[r] - kmax);
      }
      if constexpr (VEC) {
        const double2 v = *reinterpret_cast<const double2*>(wp + o);
        wreg[r][0] = v.x;
        wreg[r][1] = v.y;
      } else {
        wreg[r][0] = wp[o];
      }
    }
#pragma unroll
    for (int r = 0; r < HREGS; ++r) {
      int o = h_go[r];
      if constexpr (KEDGE) {
        const int kmax = in - E - kt * BK;
        if (h_kk[r] > kmax) o -= h_kk[r] - (kmax > 0 ? kmax : 0);
      }
      if constexpr (VEC) {
        const double2 v = *reinterpret_cast<const double2*>(hp + o);
        hreg[r][0] = v.x;
        hreg[r][1] = v.y;
      } else {
        hreg[r][0] = hp[o];
      }
    }
  };
  auto store_tiles = [&](auto BUF, int kt) {
    constexpr int buf = decltype(BUF)::value;
    double* w = sW + buf * BK * BMP;
    double* h = sH + buf * BN * BKP;
#pragma unroll
    for (int r = 0; r < WREGS; ++r) {
      if (WRAG && w_k[r] >= BK) continue;
      bool ok = true;
      if constexpr (KEDGE) ok = kt * BK + w_k[r] < in;
      if constexpr (VEC)
        *reinterpret_cast<double2*>(w + w_lds[r]) = make_double2(ok ? wreg[r][0] : 0.0, ok ? wreg[r][1] : 0.0);
      else
        w[w_lds[r]] = ok ? wreg[r][0] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < HREGS; ++r) {
      if (HRAG && h_b[r] >= BN) continue;
      bool ok = true;
      if constexpr (KEDGE) ok = kt * BK + h_kk[r] < in;
      if constexpr (VEC)
        *reinterpret_cast<double2*>(h + h_lds[r]) = make_double2(ok ? hreg[r][0] : 0.0, ok ? hreg[r][1] : 0.0);
      else
        h[h_lds[r]] = ok ? hreg[r][0] : 0.0;
    }
  };

  // Software pipeline (one barrier per 16-deep k tile, two LDS buffers, two fragment register sets).  An fp64 MFMA
  // occupies the matrix pipe for 64 cycles, so everything else is issued UNDER the MFMAs of the same wave:
  //   step 0: read frags(t,1)                                   | 16 MFMAs on frags(t,0)
  //   step 1: read frags(t,2); ds_write tile t+1 -> other buffer | 16 MFMAs on frags(t,1)
  //   step 2: read frags(t,3)                                   | 16 MFMAs on frags(t,2)
  //   barrier  (tile t+1 visible; nobody reads this buffer's tile t any more: its last reads were waited for)
  //   step 3: read frags(t+1,0); global loads of tile t+2       | 16 MFMAs on frags(t,3)
  // RAW: tile t+1 is written before the barrier of iteration t and first read after it.  WAR: the buffer of tile t
  // is next written in step 1 of iteration t+1, after every wave has passed the barrier of iteration t.
  // The loop is unrolled by two so that the buffer index, and with it every LDS offset, is an immediate.
  double fw[2][TM], fh[2][TN];
  const double* fwb = sW + wm * (BM / WM) + c + q * BMP;
  const double* fhb = sH + (wn * (BN / WN) + c) * BKP + q;
  auto read_frags = [&](auto BUF, auto S, auto SET) {
    constexpr int buf = decltype(BUF)::value, s = decltype(S)::value, set = decltype(SET)::value;
#pragma unroll
    for (int a = 0; a < TM; ++a) fw[set][a] = fwb[buf * BK * BMP + 4 * s * BMP + a * 16];
#pragma unroll
    for (int b = 0; b < TN; ++b) fh[set][b] = fhb[buf * BN * BKP + b * 16 * BKP + 4 * s];
  };
  // half of the TM*TN MFMAs of one k step (HALF = 0 / 1)
  auto mfma_half = [&](auto SET, auto HALF) {
    constexpr int set = decltype(SET)::value, half = decltype(HALF)::value;
    constexpr int lo = half == 0 ? 0 : (TM * TN) / 2, hi = half == 0 ? (TM * TN) / 2 : TM * TN;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = lo; t < hi; ++t) {
      const int a = t / TN, b = t % TN;
      acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fh[set][b], fw[set][a], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // every phase is [first half of the MFMAs][LDS / global traffic for later phases][second half]: the operands of a
  // phase were requested half a phase (>= 512 cycles) before it starts, and no wait ever sits in front of an idle pipe
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
    if (kt + 1 < nk && !(dbg & 8)) store_tiles(NBUF, kt + 1);
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
    if (!(dbg & 4)) __syncthreads();
    if (kt + 1 < nk) read_frags(NBUF, I0{}, I0{});
    if (kt + 2 < nk && !(dbg & 2)) load_tiles(kt + 2);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
  };

  load_tiles(0);
  store_tiles(I0{}, 0);
  if (nk > 1) load_tiles(1);
  __syncthreads();
  read_frags(I0{}, I0{}, I0{});
  SI_STAMP(1);
  for (int kt = 0; kt < nk; kt += 2) {
    tile_body(I0{}, I1{}, kt);
    if (kt + 1 < nk) tile_body(I1{}, I0{}, kt + 1);
  }
  SI_STAMP(2);

  // ---- epilogue: bias + activation + store.
  // A lane holds D[b = q + 4r][i = c]: stored directly that is one 8-B store per element in 128-B runs, and the
  // store ISSUE (not bandwidth) then costs as much as ~10 % of the whole 960-deep k loop.  Instead each wave transposes
  // one 16-row block at a time through its own slice of the (now idle) staging LDS and writes 16 B per lane with
  // consecutive lanes on consecutive features: every store instruction covers whole rows of the output tile.
  constexpr int WI = BM / WM;  // features per wave
  // (any WI whose 16-row block is a whole number of 64-lane sweeps: 32 / 64 / 128 features keep one feature pair per lane, 48
  //  -- the 96-row tile -- walks three of them and reloads its bias pair per chunk)
  constexpr bool WIDE = VEC && ((16 * (WI / 2)) % 64 == 0) && (WM * WN * 16 * WI <= 2 * (BK * BMP + BN * BKP));
  const int iw0 = i0 + wm * WI;
  const int64_t bw0 = b0 + wn * (BN / WN);
  auto finish = [&](double v) -> double {
    if (act == SI_ACT_RELU) return v > 0.0 ? v : 0.0;
    if (act == SI_ACT_IDENTITY) return v;
    return apply_act(v, act);
  };
  if constexpr (FUSE) {
    int gi[TM];
    double bv[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      gi[a] = iw0 + a * 16 + c;
      bv[a] = gi[a] < out ? bias[gi[a]] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = finish(acc[a][b][r] + bv[a]);
    if (Hout != nullptr) {  // block-uniform: the gradient / training forward keeps this layer's output for the reverse sweep
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t gb = bw0 + b * 16 + q + 4 * r;
            if (gi[a] < out && gb < B) Hout[gi[a] + (int64_t)out * gb] = acc[a][b][r];
          }
    }
    const int64_t slot = (int64_t)mt * WM + wm;
    for (int o = 0; o < out_last; ++o) {
      double wl[TM];
#pragma unroll
      for (int a = 0; a < TM; ++a) wl[a] = gi[a] < out ? Wlast[o + (int64_t)out_last * gi[a]] : 0.0;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double p = 0.0;
#pragma unroll
          for (int a = 0; a < TM; ++a) p = fma(acc[a][b][r], wl[a], p);
          // the 16 lanes of a 16-lane row hold the 16 features of one tile column: butterfly sum inside the row
          p += __shfl_xor(p, 8, 16);
          p += __shfl_xor(p, 4, 16);
          p += __shfl_xor(p, 2, 16);
          p += __shfl_xor(p, 1, 16);
          const int64_t gb = bw0 + b * 16 + q + 4 * r;
          if (c == 0 && gb < B) part[(slot * out_last + o) * (cb.part_ld ? cb.part_ld : B) + gb] = p;
        }
      }
    }
  } else if constexpr (WIDE) {
    constexpr int CH_ROW = WI / 2;              // 16-B chunks per row of the wave's sub-tile
    constexpr int NCH = 16 * CH_ROW / 64;       // chunks per lane per 16-row block
    double* reg = smem + wave * (16 * WI);      // wave-private: no workgroup barrier needed (LDS is in-order per wave)
    // with 64 % CH_ROW == 0 a lane keeps the same feature pair for every chunk: its bias pair is loaded once
    constexpr bool FIXCOL = 64 % CH_ROW == 0;   // the lane's feature pair is the same for every chunk
    const int col2_fixed = lane % CH_ROW;
    double2 bfix = make_double2(0.0, 0.0);
    if (FIXCOL) {
      const int gi = iw0 + 2 * col2_fixed;
      if (gi + 1 < out) bfix = *reinterpret_cast<const double2*>(bias + gi);
    }
#pragma unroll
    for (int bt = 0; bt < TN; ++bt) {
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) reg[(q + 4 * r) * WI + a * 16 + c] = acc[a][bt][r];
#pragma unroll
      for (int p = 0; p < NCH; ++p) {
        const int chunk = p * 64 + lane;
        const int row = chunk / CH_ROW, col2 = chunk % CH_ROW;
        const double2 v = *reinterpret_cast<const double2*>(reg + 2 * chunk);
        const int gi = iw0 + 2 * col2;
        const int64_t gb = bw0 + bt * 16 + row;
        double2 bv = bfix;
        if (!FIXCOL) bv = (gi + 1 < out) ? *reinterpret_cast<const double2*>(bias + gi) : make_double2(0.0, 0.0);
        if (gi + 1 < out && gb < B && !(dbg & 16))
          *reinterpret_cast<double2*>(Hout + gi + (int64_t)out * gb) = make_double2(finish(v.x + bv.x), finish(v.y + bv.y));
      }
    }
  } else {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const int gi = iw0 + a * 16 + c;
      const double bv = (gi < out) ? bias[gi] : 0.0;
#pragma unroll
      for (int b = 0; b < TN; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gb = bw0 + b * 16 + q + 4 * r;
          if (gi < out && gb < B) Hout[gi + (int64_t)out * gb] = finish(acc[a][b][r] + bv);
        }
      }
    }
  }
  SI_STAMP(3);
}

struct FuseArgs {
  const double* Wlast = nullptr;  // out_last x out, column-major (the last layer's weights inside the flat vector)
  int out_last = 0;
  double* part = nullptr;         // [slots][out_last][B]
  ChainBatch cb;                  // chain slots in grid.y
};

// The last round of workgroups (measured in round 3 and left alone).  cfg2's layers are 7820 tiles on 512 slots (two
// 8-wave workgroups per CU) = 15 rounds + 140 tiles: the last round keeps 140 CUs busy for a whole tile time and 116 idle
// -- 1.7 % of the kernel at 15.27 rounds, 2.6 % at 10.2 (tools/tail_quant_probe.py).  Cutting that round into half-width
// tiles was built twice, bit-identical both times (same k order, same per-wave feature slots of the fused head):
//   * both widths in ONE kernel through a shared tile function: +2.2 % at 10.2 rounds, +0.3 % at cfg2 against itself --
//     but the combined kernel's k loop came out with a different block structure and ran 1.8 % SLOWER at cfg2 than the
//     kernel below (A/B against the library of the commit before, same box);
//   * a second launch of the same template at BN = 64 on pointers shifted to the first column behind the last full
//     round: +0.4 % at 10.2 rounds, 0.8 % slower at cfg2 (270 half tiles still pair up on 14 CUs, and the extra kernel
//     boundary costs more than the idle CUs did).
// profiles/r03_tail_round_ab.log.
template <int BM, int BN, int WM, int WN, int MINW, bool VEC, bool KEDGE, bool FUSE>
static void launch_dense_inst(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout,
                              int32_t out, int32_t in, int64_t B, int32_t act, const FuseArgs& fa) {
  size_t lds = 2 * (16 * (BM + 16) + BN * 18) * sizeof(double);
#ifdef SI_GEMM_DEBUG_KNOB
  if (si_gemm_lds_floor > lds) lds = si_gemm_lds_floor;  // harness: force fewer workgroups per CU
#endif
  constexpr int NT = 64 * WM * WN;
  const int nMt = (out + BM - 1) / BM;
  const int64_t nNt = (B + BN - 1) / BN;
  const int64_t groups = (nNt + 7) / 8;  // batch panels per XCD lane
  const int64_t grid = groups * nMt * 8;
  auto kern = dense_f64_kernel<BM, BN, WM, WN, MINW, VEC, KEDGE, FUSE>;
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)fa.cb.n), dim3(NT), lds, st, W, bias, Hin, Hout, (int)out,
                     (int)in, B, (int)act, nMt, nNt, fa.Wlast, fa.out_last, fa.part, fa.cb);
}

template <int BM, int BN, int WM, int WN, int MINW, bool FUSE = false>
void launch_dense_cfg(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout,
                      int32_t out, int32_t in, int64_t B, int32_t act, const FuseArgs& fa = FuseArgs()) {
  // 16-B staging accesses need even strides and 16-B aligned bases (W sits at an arbitrary offset of the flat
  // weight vector: layer 3 of cfg2 starts at an odd element); a ragged k edge (in % 16 != 0) needs zero-fill
  const bool vec = (out % 2 == 0) && (in % 2 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15u) == 0) &&
                   ((reinterpret_cast<uintptr_t>(Hin) & 15u) == 0) &&
                   (FUSE || (((reinterpret_cast<uintptr_t>(Hout) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(bias) & 15u) == 0))) &&
                   (((fa.cb.w | fa.cb.hin | fa.cb.hout) & 1) == 0);
  const bool kedge = (in % 16) != 0;
  if (vec && !kedge)
    launch_dense_inst<BM, BN, WM, WN, MINW, true, false, FUSE>(st, W, bias, Hin, Hout, out, in, B, act, fa);
  else if (vec)
    launch_dense_inst<BM, BN, WM, WN, MINW, true, true, FUSE>(st, W, bias, Hin, Hout, out, in, B, act, fa);
  else if (!kedge)
    launch_dense_inst<BM, BN, WM, WN, MINW, false, false, FUSE>(st, W, bias, Hin, Hout, out, in, B, act, fa);
  else
    launch_dense_inst<BM, BN, WM, WN, MINW, false, true, FUSE>(st, W, bias, Hin, Hout, out, in, B, act, fa);
}

#ifndef SI_GEMM_NO_DISPATCH
// Tile choice (tools/gemm_bench.hip sweep on MI355X, layer 960x960xB=1e5, fp64): two 8-wave workgroups per CU
// (4 waves per SIMD) beat every 2-waves-per-SIMD shape -- 96x128: 68.2, 64x128: 67.2, 128x128: 65.1 TFLOP/s against
// 56-60 for 4-wave 128x128 / 192x128 tiles -- because the prologue / epilogue of one workgroup hides under three other
// waves' MFMAs.  The feature tile BM is the one that pads `out` least (960 = 10 x 96).
static int pick_bm(int32_t out) {
  if (out <= 32) return 32;
  auto padded = [&](int bm) { return (out + bm - 1) / bm * bm; };
  const int p96 = padded(96), p128 = padded(128), p64 = padded(64);
  if (p96 <= p128 && p96 <= p64) return 96;
  return p128 <= p64 ? 128 : 64;
}

int dense_fused_slot_feats(int32_t out);
int dense_fused_slots(int32_t out);

void launch_dense_f64(hipStream_t st, const double* W, const double* bias, const double* Hin,
                      double* Hout, int32_t out, int32_t in, int64_t B, int32_t act, const ChainBatch& cb) {
  if (act_is_extra(act)) {   // leakyrelu / elu / softplus / selu: identity in the GEMM, one elementwise pass behind it
    launch_dense_f64(st, W, bias, Hin, Hout, out, in, B, SI_ACT_IDENTITY, cb);
    for (int s = 0; s < cb.n; ++s) launch_act_inplace(st, Hout + (int64_t)s * cb.hout, (int64_t)out * B, act);
    return;
  }
  if (dense_small_applies(out, in, B, cb.n, pick_bm(out))) {   // a handful of big tiles: all latency (kernels_gemm_small.hip, same bits)
    launch_dense_small_f64(st, W, bias, Hin, Hout, out, in, B, act, dense_fused_slot_feats(out), cb);
    return;
  }
  if (cb.n <= 1 && dense_panel_applies(W, out, in, B, act) && launch_dense_f64_panel(st, W, bias, Hin, Hout, out, in, B, act))
    return;   // a wide first layer (kernels_gemm_panel.hip, same bits)
  FuseArgs fa;
  fa.cb = cb;
  switch (pick_bm(out)) {
    case 32: launch_dense_cfg<32, 128, 1, 4, 2>(st, W, bias, Hin, Hout, out, in, B, act, fa); break;
    case 96: launch_dense_cfg<96, 128, 2, 4, 4>(st, W, bias, Hin, Hout, out, in, B, act, fa); break;
    case 128: launch_dense_cfg<128, 128, 2, 4, 4>(st, W, bias, Hin, Hout, out, in, B, act, fa); break;
    default: launch_dense_cfg<64, 128, 2, 4, 4>(st, W, bias, Hin, Hout, out, in, B, act, fa); break;
  }
}

// number of feature slots (partials per (o, b)) the fused kernel writes for a layer of width `out`
int dense_fused_slots(int32_t out) {
  const int bm = pick_bm(out);
  const int wm = bm == 32 ? 1 : 2;
  return (out + bm - 1) / bm * wm;
}

// features per slot (= per wave of a feature tile): the unit the fused head sums in one fma chain
int dense_fused_slot_feats(int32_t out) {
  const int bm = pick_bm(out);
  return bm == 32 ? 32 : bm / 2;
}

void launch_dense_f64_fused(hipStream_t st, const double* W, const double* bias, const double* Hin, int32_t out,
                            int32_t in, int64_t B, int32_t act, const double* Wlast, int32_t out_last, double* part,
                            const ChainBatch& cb, double* Hkeep) {
  if (dense_small_applies(out, in, B, cb.n, pick_bm(out))) {
    launch_dense_small_f64_fused(st, W, bias, Hin, out, in, B, act, dense_fused_slot_feats(out), dense_fused_slots(out), Wlast, out_last,
                                 part, cb, Hkeep);
    return;
  }
  FuseArgs fa;
  fa.Wlast = Wlast;
  fa.out_last = out_last;
  fa.part = part;
  fa.cb = cb;
  switch (pick_bm(out)) {
    case 32: launch_dense_cfg<32, 128, 1, 4, 2, true>(st, W, bias, Hin, Hkeep, out, in, B, act, fa); break;
    case 96: launch_dense_cfg<96, 128, 2, 4, 4, true>(st, W, bias, Hin, Hkeep, out, in, B, act, fa); break;
    case 128: launch_dense_cfg<128, 128, 2, 4, 4, true>(st, W, bias, Hin, Hkeep, out, in, B, act, fa); break;
    default: launch_dense_cfg<64, 128, 2, 4, 4, true>(st, W, bias, Hin, Hkeep, out, in, B, act, fa); break;
  }
}

// ------------------------------------------------------------------------------------------------
// Tail of the fused path: yhat[o][b] = act_last(sum_slots part + b_last[o]);  block partials of (y - yhat)^2.
// Fixed summation order => bit-reproducible.  Replaces the last Dense layer + the SSE pass of
// reference src/space_inference.jl:94.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tail_sse_kernel(const double* __restrict__ part, int slots, int out_last,
                                                       int64_t B, const double* __restrict__ bias_last, int act_last,
                                                       const double* __restrict__ Y, double* __restrict__ yhat,
                                                       double* __restrict__ blockpart, ChainBatch cb) {
  __shared__ double red[4];
  double accv = 0.0;
  const int64_t d = (int64_t)out_last * B;
  {  // chain slot in grid.y: own partial products, last-layer bias, output and block partials; Y is shared
    const int64_t ch = blockIdx.y;
    part += ch * cb.part;
    bias_last += ch * cb.w;
    if (yhat) yhat += ch * d;
    blockpart += ch * gridDim.x;
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < d; idx += stride) {
    const int o = (int)(idx % out_last);
    const int64_t b = idx / out_last;
    double s = 0.0;
#pragma unroll 4   // the loads of four slots in flight together (the additions keep their order: bit-identical)
    for (int sl = 0; sl < slots; ++sl) s += part[((int64_t)sl * out_last + o) * B + b];
    double v = s + bias_last[o];
    if (act_last == SI_ACT_RELU)
      v = v > 0.0 ? v : 0.0;
    else if (act_last != SI_ACT_IDENTITY)
      v = act_full(v, act_last);
    if (yhat) yhat[idx] = v;
    const double r = Y[idx] - v;
    accv += r * r;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) accv += __shfl_down(accv, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = accv;
  __syncthreads();
  if (threadIdx.x == 0) blockpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

void launch_tail_sse(hipStream_t st, const double* part, int slots, int out_last, int64_t B, const double* bias_last,
                     int act_last, const double* Y, double* yhat, double* blockpart, int nblocks, const ChainBatch& cb) {
  hipLaunchKernelGGL(tail_sse_kernel, dim3(nblocks, cb.n), dim3(256), 0, st, part, slots, out_last, B, bias_last,
                     act_last, Y, yhat, blockpart, cb);
}
#endif

}  // namespace si
