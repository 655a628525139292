// Backward pass of the Dense chain in fp64: the gradient of the log-density (or of the mse
// training loss) with respect to the flat weight vector, and its pull-back to the subspace, grad_z = P' grad_w.
// The reference obtains d lp / d z by pushing M-wide ForwardDiff duals through the whole network
// (src/space_inference.jl:107 `getbackend(backend).gradient(density, theta)`); one reverse sweep gives the same
// numbers with 2x the forward flops instead of Mx.
//
//   Delta_L     = g .* act_L'(H_L)                              delta_out_kernel
//   dW_l        = Delta_l * H_{l-1}'      (out x in, K = B)     gemm_f64_kernel<A m-fast, B n-fast, split-K> + reduce
//   db_l        = rowsum(Delta_l)                               rowsum kernels
//   Delta_{l-1} = (W_l' * Delta_l) .* act_{l-1}'(H_{l-1})       gemm_f64_kernel<A k-fast, B k-fast, EPI_DACT>
//   grad_z      = P' * grad_w                                   pt_g kernels
//
// gemm_f64_kernel is the forward kernel's pipeline (kernels_gemm.hip: 8 waves, 16-deep k tiles, two LDS buffers, one
// barrier per tile, MFMAs split around the LDS/global traffic, v_mfma_f64_16x16x4_f64) with the operand staging made
// generic over the two storage orders an operand can have:
//   layout 0 "row-fast": X[r + ld*k]  -> LDS [k][R+16]      layout 1 "k-fast": X[k + ld*r] -> LDS [r][18]
// Both LDS images give conflict-free ds_read_b64 operand reads (same bank arithmetic as the forward kernel).
#include <algorithm>

#include "gemm_pipeline.h"

namespace si {

enum { EPI_DACT = 1, EPI_RAW = 2 };
constexpr int RS_CHUNKS = 64;   // column chunks of the row-sum kernels (db = rowsum(Delta))

#ifdef SI_BWD_DEBUG_KNOB
__device__ int si_bwd_dbg = 0;  // harness only: 1 = every block reads split 0 / tile (0,0) (cache-hot), 2 = no global loads in the k loop
#endif


__device__ __forceinline__ double dact_from_output(double h, int act) {
  switch (act) {
    case SI_ACT_RELU: return h > 0.0 ? 1.0 : 0.0;       // relu'(z) = [z > 0] = [relu(z) > 0]
    case SI_ACT_TANH: return 1.0 - h * h;
    case SI_ACT_SIGMOID: return h * (1.0 - h);
    default: return 1.0;   // (the later activations never reach the GEMM epilogue: launch_backward_data)
  }
}

// C[m + ldc*n] (+ epilogue) = sum_k A(m,k) * B(k,n) over k in [k0, k0+klen) of split blockIdx.y
template <int BM, int BN, int WM, int WN, int MINW, int ALAY, int BLAY, bool VEC, int EPI>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_f64_kernel(
    const double* __restrict__ A, int64_t lda, const double* __restrict__ Bm, int64_t ldb, double* __restrict__ C,
    int64_t ldc, int Mrows, int64_t Ncols, int64_t Kdim, int64_t ksplit, const double* __restrict__ aux, int act, int nMt,
    int64_t nNt) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  using SA = Stager<BM, ALAY, NT, VEC>;
  using SB = Stager<BN, BLAY, NT, VEC>;
  extern __shared__ double smem[];

  // nNt >= 8: XCD-grouped map of the forward kernel (the nMt row tiles of one column panel share an XCD's L2).
  // nNt < 8 (weight gradients of narrow layers): that map would leave XCDs empty (measured: a 960x128 dW on ONE XCD,
  // 7x slower), so the tiles are numbered plainly and the split index spreads the blocks over the XCDs.
  // (A map in which every XCD owns whole k-ranges of a split-K launch -- each element of both operands enters exactly
  // one L2 -- was measured too: same speed, the kernel is not bound by L2 misses; not kept.)
  const int64_t bid = blockIdx.x;
  int mt;
  int64_t nt;
  const int64_t split = blockIdx.y;
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
  const int64_t k0 = split * ksplit;
  int64_t klen = Kdim - k0;
  if (klen > ksplit) klen = ksplit;
  if (klen <= 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int m0 = mt * BM;
  const int64_t n0 = nt * BN;
  d4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

  SA sa;
  SB sb;
#ifdef SI_BWD_DEBUG_KNOB
  const int dbg = si_bwd_dbg;
  sa.init(A, lda, (dbg & 1) ? 0 : m0, Mrows, (dbg & 1) ? 0 : k0, tid);
  sb.init(Bm, ldb, (dbg & 1) ? 0 : n0, Ncols, (dbg & 1) ? 0 : k0, tid);
#else
  constexpr int dbg = 0;
  sa.init(A, lda, m0, Mrows, k0, tid);
  sb.init(Bm, ldb, n0, Ncols, k0, tid);
#endif
  const int nk = (int)((klen + 15) / 16);

  gemm_mainloop<BM, BN, WM, WN>(sa, sb, smem, nk, klen, wm, wn, lane, acc, dbg);

  // epilogue: optional multiplication with act'(aux) (EPI_DACT), split partials behind one another (EPI_RAW)
  double* Cout = C + (EPI == EPI_RAW ? split * ldc * Ncols : 0);
  gemm_epilogue<BM, BN, WM, WN, VEC, 2 * (SA::LDS_ELEMS + SB::LDS_ELEMS)>(
      acc, smem, Cout, ldc, m0, n0, Mrows, Ncols, wm, wn, lane, wave, [&](double v, int64_t off, int) {
        if constexpr (EPI == EPI_DACT) v *= dact_from_output(aux[off], act);
        return v;
      });
}

template <int BM, int BN, int ALAY, int BLAY, int EPI>
static void launch_gemm(hipStream_t st, const double* A, int64_t lda, const double* Bm, int64_t ldb, double* C, int64_t ldc,
                        int Mrows, int64_t Ncols, int64_t Kdim, int nsplit, int64_t ksplit, const double* aux, int act) {
  constexpr int WM = 2, WN = 4, NT = 512;
  using SA = Stager<BM, ALAY, NT, false>;
  using SB = Stager<BN, BLAY, NT, false>;
  constexpr size_t lds = 2 * (SA::LDS_ELEMS + SB::LDS_ELEMS) * sizeof(double);
  const int nMt = (Mrows + BM - 1) / BM;
  const int64_t nNt = (Ncols + BN - 1) / BN;
  const int64_t grid = nNt >= 8 ? (nNt + 7) / 8 * nMt * 8 : nNt * nMt;
  // 16-B staging: pairs run along the fast index of each operand, so that extent and the leading dimensions must be
  // even and the bases 16-B aligned; the k-fast pairs also need every split to start at an even k
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  const bool vec = al(A) && al(Bm) && al(C) && (aux == nullptr || al(aux)) && lda % 2 == 0 && ldb % 2 == 0 &&
                   ldc % 2 == 0 && Mrows % 2 == 0 && (ALAY == 0 || Kdim % 2 == 0) &&
                   (BLAY == 0 ? Ncols % 2 == 0 : Kdim % 2 == 0) && ksplit % 2 == 0;
  if (vec) {
    auto kern = gemm_f64_kernel<BM, BN, WM, WN, 4, ALAY, BLAY, true, EPI>;
    static LdsOptIn optin;
    optin.ensure(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)nsplit), dim3(NT), lds, st, A, lda, Bm, ldb, C, ldc, Mrows,
                       Ncols, Kdim, ksplit, aux, act, nMt, nNt);
  } else {
    auto kern = gemm_f64_kernel<BM, BN, WM, WN, 4, ALAY, BLAY, false, EPI>;
    static LdsOptIn optin;
    optin.ensure(reinterpret_cast<const void*>(kern), lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)nsplit), dim3(NT), lds, st, A, lda, Bm, ldb, C, ldc, Mrows,
                       Ncols, Kdim, ksplit, aux, act, nMt, nNt);
  }
}

// ---- weight gradient with LDS-DMA staging (round 4) -------------------------------------------------------------
// dW = Delta * Hprev' contracts over the BATCH: both operands are "row-fast" (X[r + ld*k], one k = one observation = R
// contiguous doubles), the k loop is ~1000 tiles long and nothing but MFMAs happens in it.  The register-staged kernel above
// needs 20 staging VGPRs, so a 3 x 3 tile layout per wave (96 x 192 per workgroup: no padded columns for widths like 960,
// 9 MFMAs per 6 fragment reads) spills (measured: 100 VGPRs spilled at the 128 budget of 4 waves per SIMD).  Here a tile's
// 16 x (BM + BN) doubles go global -> LDS directly (global_load_lds_dwordx4, 16 B per lane, linear destination), two stages:
//   tile kt:  s_waitcnt vmcnt(0); s_barrier;  issue the DMA of tile kt+1 into the other stage;  36 MFMAs per wave on tile kt
// so a DMA has a whole tile's compute time to land.  LDS image of one operand: [k][R] doubles UNPADDED; the operand read of
// lane (q, c) is element [4s + q][blk*16 + c], and rows of R = 96 / 128 / 192 doubles are a multiple of 64 banks apart, so rows
// q and q+1 would collide: odd k rows are stored with neighbouring 16-double blocks exchanged (block ^ 1) -- done on the DMA
// SOURCE address, the destination stays linear -- and the 32 lanes of a ds_read_b64 pass cover all 64 banks.
// Needs: whole k tiles in every split (B % 16 == 0), even out / in, 16-byte aligned operands; anything else takes the
// register-staged kernel.
typedef __attribute__((address_space(3))) void* lds_void_ptr_d;

template <int N, int I = 0, class F>
__device__ __forceinline__ void bwd_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    bwd_static_for<N, I + 1>(f);
  }
}

// (body and kernel apart: a __global__ template with device builtins inside lambdas loses its host stub)
template <int BM, int BN>
__device__ __forceinline__ void dw_f64_dma_body(const double* __restrict__ A, int64_t lda, const double* __restrict__ Bm,
                                                int64_t ldb, double* __restrict__ C, int64_t ldc, int Mrows, int64_t Ncols,
                                                int64_t Kdim, int64_t ksplit, int nMt, int nNt, int nsplit,
                                                double* __restrict__ dbpart, double* smem) {
  constexpr int WM = 2, WN = 4, NWAVES = 8;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int STAGE = 16 * (BM + BN);              // doubles per stage
  constexpr int NA = 16 * BM / 128, NBI = 16 * BN / 128;   // 1 KiB DMA instructions per tile: A part, B part
  static_assert(BM % 32 == 0 && BN % 64 == 0 && (BM / 16) % 2 == 0 && (BN / 16) % 2 == 0, "tile shape");

  // Workgroup -> (tile, split): the launch is ONE round of workgroups, dealt to the 8 XCDs round-robin by the dispatcher.  XCD x
  // takes the contiguous range [x * per_xcd, (x+1) * per_xcd) of the split-major work list, i.e. one or two whole k ranges with
  // all their tiles: the tiles of a split walk the same columns of Delta and H together and find them in that XCD's L2.
  // (Tiles numbered plainly over blockIdx: FETCH_SIZE 10.7 GB per launch at cfg2 -- every tile fetched its operands past the
  // L2 -- against 1.5 GB of operands.)
  const int tiles = nMt * nNt;
  const int total = tiles * nsplit;
  const int per_xcd = (total + 7) >> 3;
  const int work = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per_xcd || work >= total) return;   // uniform per workgroup, before any barrier
  const int64_t split = work / tiles;
  const int tile = work % tiles;
  const int mt = tile % nMt;
  const int64_t nt = tile / nMt;
  const int64_t k0 = split * ksplit;
  int64_t klen = Kdim - k0;
  if (klen > ksplit) klen = ksplit;
  if (klen <= 0) return;
  const int nk = (int)(klen / 16);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int m0 = mt * BM;
  const int64_t n0 = nt * BN;
  const int q = lane >> 4, c = lane & 15;

  // DMA plan of this wave: the A image takes NA instructions (1 KiB each), the B image NBI; slot s of an operand is instruction
  // wave + 8 s of that operand, so which operand a slot belongs to is known at compile time and only the last A slot can be
  // partial (BM = 96: 12 instructions = one slot of all waves + one of waves 0-3).  A lane keeps one 32-bit BYTE offset per slot
  // (its element inside the tile's 16 columns); the tile's base address is wave-uniform: the scalar-base form of the load.
  constexpr int SA_N = (NA + NWAVES - 1) / NWAVES, SB_N = (NBI + NWAVES - 1) / NWAVES;
  static_assert(NBI % NWAVES == 0, "the B image is whole slots");
  uint32_t aoff[SA_N], boff[SB_N];
#pragma unroll
  for (int s = 0; s < SA_N; ++s) {
    const int id = wave + NWAVES * s;
    const int d = (id < NA ? id : 0) * 64 + lane;   // 16-byte pair inside the A image: row k = d / (BM/2)
    const int k = d / (BM / 2), pp = d % (BM / 2);
    int m = m0 + 2 * (pp ^ ((k & 1) << 3));
    if (m > Mrows - 2) m = Mrows - 2;               // clamped rows only feed outputs that are never stored
    aoff[s] = (uint32_t)(m + (int)lda * k) * 8u;
  }
#pragma unroll
  for (int s = 0; s < SB_N; ++s) {
    const int d = (wave + NWAVES * s) * 64 + lane;
    const int k = d / (BN / 2), pp = d % (BN / 2);
    int64_t n = n0 + 2 * (pp ^ ((k & 1) << 3));
    if (n > Ncols - 2) n = Ncols - 2;
    boff[s] = (uint32_t)((int)n + (int)ldb * k) * 8u;
  }
  auto issue = [&](int kt, int buf) {
    double* dst = smem + buf * STAGE;
    const char* ta = reinterpret_cast<const char*>(A + lda * (k0 + 16 * (int64_t)kt));
    const char* tb = reinterpret_cast<const char*>(Bm + ldb * (k0 + 16 * (int64_t)kt));
    bwd_static_for<SA_N>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      if ((s + 1) * NWAVES <= NA || wave + NWAVES * s < NA)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(ta + aoff[s]),
                                         (lds_void_ptr_d)(dst + (wave + NWAVES * s) * 128), 16, 0, 0);
    });
    bwd_static_for<SB_N>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(tb + boff[s]),
                                       (lds_void_ptr_d)(dst + 16 * BM + (wave + NWAVES * s) * 128), 16, 0, 0);
    });
  };

  d4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};

  // fragment offsets (doubles inside a stage) of k step 0; step s adds 4 s BM / 4 s BN
  int offA[TM], offB[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a) offA[a] = q * BM + (((wm * TM + a) ^ (q & 1)) << 4) + c;
#pragma unroll
  for (int b = 0; b < TN; ++b) offB[b] = 16 * BM + q * BN + (((wn * TN + b) ^ (q & 1)) << 4) + c;

  // db = rowsum(Delta) rides along (dbpart != nullptr): the waves of the first column tile that hold distinct rows (wn == 0) add up
  // their A fragments -- every element of the tile's rows passes through them exactly once -- in the fixed order of the k loop
  const bool do_rs = dbpart != nullptr && nt == 0 && wn == 0;   // wave-uniform
  double rs[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) rs[a] = 0.0;

  // (two copies of the loop, chosen once: as one loop with a wave-uniform `if` the row sums were compiled to unconditional
  //  adds + selects in every wave)
  auto mainloop = [&](auto RS) {
    constexpr bool with_rs = decltype(RS)::value;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // every wave's pieces of tile kt are in; every wave is done reading tile kt-1
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      const double* st = smem + (kt & 1) * STAGE;
      // (one fragment set: the other three waves of the SIMD cover the LDS latency; a second, software-pipelined set costs 12
      //  registers and pushed the kernel over its 128-register budget -- spills inside the loop)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        double fa[TM], fb[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[a] = st[offA[a] + 4 * s * BM];
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[b] = st[offB[b] + 4 * s * BN];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0);
        if constexpr (with_rs) {
#pragma unroll
          for (int a = 0; a < TM; ++a) rs[a] += fa[a];
        }
      }
    }
  };
  if (do_rs) mainloop(std::true_type{});
  else mainloop(std::false_type{});
  if (do_rs) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      double t = rs[a];
      t += __shfl_xor(t, 16);   // the four q groups hold k = 4s + q of the same row
      t += __shfl_xor(t, 32);
      const int gm = m0 + wm * (BM / WM) + a * 16 + c;
      if (q == 0 && gm < Mrows) dbpart[split * Mrows + gm] = t;
    }
  }

  double* Cout = C + split * ldc * Ncols;
  gemm_epilogue<BM, BN, WM, WN, true, 2 * STAGE>(acc, smem, Cout, ldc, m0, n0, Mrows, Ncols, wm, wn, lane, wave,
                                                 [&](double v, int64_t, int) { return v; });
}

template <int BM, int BN>
__global__ __launch_bounds__(512, 4) void dw_f64_dma_kernel(const double* __restrict__ A, int64_t lda,
                                                            const double* __restrict__ Bm, int64_t ldb, double* __restrict__ C,
                                                            int64_t ldc, int Mrows, int64_t Ncols, int64_t Kdim, int64_t ksplit,
                                                            int nMt, int nNt, int nsplit, double* __restrict__ dbpart) {
  extern __shared__ double smem_dw[];
  dw_f64_dma_body<BM, BN>(A, lda, Bm, ldb, C, ldc, Mrows, Ncols, Kdim, ksplit, nMt, nNt, nsplit, dbpart, smem_dw);
}

template <int BM, int BN>
static void launch_dw_dma(hipStream_t st, const double* Delta, const double* Hprev, double* part, int32_t out, int32_t in,
                          int64_t B, int nsplit, int64_t ks, double* dbpart) {
  constexpr size_t lds = 2 * 16 * (BM + BN) * sizeof(double);
  const int nMt = (out + BM - 1) / BM;
  const int64_t nNt = (in + BN - 1) / BN;
  auto kern = dw_f64_dma_kernel<BM, BN>;
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(kern), lds);
  const int64_t total = (int64_t)nMt * nNt * nsplit;
  hipLaunchKernelGGL(kern, dim3((unsigned)((total + 7) / 8 * 8)), dim3(512), lds, st, Delta, (int64_t)out, Hprev,
                     (int64_t)in, part, (int64_t)out, out, (int64_t)in, B, ks, nMt, (int)nNt, nsplit, dbpart);
}

static bool dw_dma_ok(const double* Delta, const double* Hprev, const double* part, int32_t out, int32_t in, int64_t B) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  return B % 16 == 0 && B >= 16 && out % 2 == 0 && in % 2 == 0 && out >= 2 && in >= 2 && al(Delta) && al(Hprev) && al(part);
}

// row tile that pads `rows` least (same rule as the forward kernel)
static int pick_bm_bwd(int32_t rows) {
  if (rows <= 64) return 64;
  auto padded = [&](int bm) { return (rows + bm - 1) / bm * bm; };
  const int p96 = padded(96), p128 = padded(128), p64 = padded(64);
  if (p96 <= p128 && p96 <= p64) return 96;
  return p128 <= p64 ? 128 : 64;
}

// Delta_prev[in x B] = (W' * Delta) .* act_prev'(Hprev);  W is out x in (column-major) inside the flat vector
void launch_backward_data(hipStream_t st, const double* W, const double* Delta, const double* Hprev, double* DeltaPrev,
                          int32_t out, int32_t in, int64_t B, int32_t act_prev) {
  if (act_is_extra(act_prev)) {   // W' Delta by the GEMM, the act' factor of a later activation by one elementwise pass
    launch_backward_data(st, W, Delta, Hprev, DeltaPrev, out, in, B, SI_ACT_IDENTITY);
    launch_mul_dact(st, DeltaPrev, Hprev, (int64_t)in * B, act_prev, DeltaPrev);
    return;
  }
  // A(m = in idx, k = out idx) = W[k + out*m]: k-fast;  B(k, n = b) = Delta[k + out*n]: k-fast
  const int64_t ks = ((int64_t)out + 15) / 16 * 16;
  switch (pick_bm_bwd(in)) {
    case 96: launch_gemm<96, 128, 1, 1, EPI_DACT>(st, W, out, Delta, out, DeltaPrev, in, in, B, out, 1, ks, Hprev, act_prev); break;
    case 128: launch_gemm<128, 128, 1, 1, EPI_DACT>(st, W, out, Delta, out, DeltaPrev, in, in, B, out, 1, ks, Hprev, act_prev); break;
    default: launch_gemm<64, 128, 1, 1, EPI_DACT>(st, W, out, Delta, out, DeltaPrev, in, in, B, out, 1, ks, Hprev, act_prev); break;
  }
}

// ---- weight gradient  dW[out x in] = Delta * Hprev'  (K = B: a handful of output tiles, a very long contraction)
// Plan of the split-K launch: the row tile and the split count that waste least -- padding of `out` to the tile, and
// unfilled slots of the ONE round of 2 x CU workgroups (two 8-wave workgroups per CU) the launch is sized for.
// cfg2 layer 2: 96-row tiles, 80 tiles x 6 splits = 480 of 512 slots.
// Measured and not kept (round 3, tools/bwd_bench.hip, profiles/r03_dw_streamk.log): a "stream-K" cut -- the k tiles of
// all output tiles as one sequence dealt out in 512 equal contiguous ranges, a workgroup finishing one tile and going
// on in the next -- fills every slot but is SLOWER (3.46 against 3.32 ms): neighbouring workgroups then walk different
// k ranges of the SAME tile and share no operand data, where the 80 tiles of one split walk the same k range of Delta
// and H together and find it in the L2 / Infinity Cache.
struct DwPlan {
  int bm, bn;
  int nsplit;
  int64_t ks;
};
// (row tile, column tile) candidates.  96 x 192 (3 x 3 MFMA tiles per wave, 80 KiB of staging LDS: exactly two workgroups
// per CU) exists for widths like cfg2's 960 = 7.5 x 128: 50 tiles x 10 splits = 500 of 512 slots and no padded columns,
// where 96 x 128 computes 960 x 1024 on 480 slots.
static constexpr int DW_NCAND = 4;
static constexpr int DW_BM[DW_NCAND] = {96, 128, 64, 96};
static constexpr int DW_BN[DW_NCAND] = {128, 128, 128, 192};
static DwPlan plan_dw(int32_t out, int32_t in, int64_t B, int num_cu, bool dma = true) {
  const int64_t slots = (int64_t)num_cu * 2;
  const int64_t maxsplit = (B + 255) / 256;
  DwPlan best{96, 128, 1, B};
  double best_score = -1.0;
  for (int c = 0; c < DW_NCAND; ++c) {
    const int bm = DW_BM[c], bn = DW_BN[c];
    if (bn != 128 && !dma) continue;   // the register-staged kernel has 128-column tiles only
    const int64_t pm = (out + bm - 1) / bm, pn = (in + bn - 1) / bn;
    const int64_t tiles = pm * pn;
    int64_t ns = slots / tiles;
    if (ns > maxsplit) ns = maxsplit;
    if (ns < 1) ns = 1;
    const double useful = ((double)out * in) / ((double)pm * bm * (double)pn * bn);
    const double fill = (double)(tiles * ns) / (double)((tiles * ns + slots - 1) / slots * slots);
    if (useful * fill > best_score * 1.0001) {
      best_score = useful * fill;
      best.bm = bm;
      best.bn = bn;
      best.nsplit = (int)ns;
    }
  }
#ifdef SI_BWD_DEBUG_KNOB
  if (const char* e = getenv("SI_BWD_BM")) best.bm = atoi(e);
  if (const char* e = getenv("SI_BWD_BN")) best.bn = atoi(e);
  if (const char* e = getenv("SI_BWD_NSPLIT")) best.nsplit = atoi(e);
#endif
  // k-range per split rounded up to whole 16-deep tiles; only splits that hold columns are launched
  best.ks = ((B + best.nsplit - 1) / best.nsplit + 15) / 16 * 16;
  best.nsplit = (int)((B + best.ks - 1) / best.ks);
  return best;
}

// doubles of scratch launch_backward_weight needs for this layer at ANY batch of up to B columns.  The plan of a smaller
// batch can hold MORE splits than the plan of B itself (the k range of a split is rounded up to whole 16-deep tiles, so
// the split count is not monotonic in the batch: out = 95, in = 33 takes 243 splits at B = 66 003 and 256 at 65 536) and
// can pick another row tile -- a training step on a last, smaller batch of an epoch wrote behind a buffer sized from
// the plan of the full batch (found by tools/guard_fuzz.py under the guard-page allocator, round 3).  The bound: no plan
// exceeds min(slots / tiles, ceil(B / 256)) splits for its tile shape.
size_t backward_weight_part_elems(int32_t out, int32_t in, int64_t B, int num_cu) {
  const int64_t slots = (int64_t)num_cu * 2, maxsplit = (B + 255) / 256;
  int64_t worst = 1;
  for (int c = 0; c < DW_NCAND; ++c) {
    const int64_t tiles = (int64_t)((out + DW_BM[c] - 1) / DW_BM[c]) * ((in + DW_BN[c] - 1) / DW_BN[c]);
    worst = std::max(worst, std::max<int64_t>(1, std::min(slots / tiles, maxsplit)));
  }
  // + the row-sum partials of the fused db (one row of `out` per split) or the scratch of launch_rowsum
  return (size_t)worst * out * in + (size_t)std::max<int64_t>(worst, RS_CHUNKS) * out;
}

// dW[out x in] = Delta * Hprev' into dW: split-K GEMM into `part`, then the splits added in fixed order
// db != nullptr: db = rowsum(Delta) as well (fused into the LDS-DMA kernel where that runs, else by the row-sum kernels)
void launch_backward_weight(hipStream_t st, const double* Delta, const double* Hprev, double* part, int32_t out,
                            int32_t in, int64_t B, int num_cu, double* dW, double* db) {
  bool dma = dw_dma_ok(Delta, Hprev, part, out, in, B);
#ifdef SI_BWD_DEBUG_KNOB
  if (getenv("SI_BWD_NODMA")) dma = false;
#endif
  const DwPlan p = plan_dw(out, in, B, num_cu, dma);
  // A(m = out idx, k = b) = Delta[m + out*k]: row-fast;  B(k = b, n = in idx) = Hprev[n + in*k]: row-fast
  double* extra = part + (size_t)p.nsplit * out * in;   // behind the dW partials: row-sum partials / launch_rowsum's scratch
  if (dma && p.bm == 96 && (p.bn == 192 || p.bn == 128)) {
    if (p.bn == 192) launch_dw_dma<96, 192>(st, Delta, Hprev, part, out, in, B, p.nsplit, p.ks, db ? extra : nullptr);
    else launch_dw_dma<96, 128>(st, Delta, Hprev, part, out, in, B, p.nsplit, p.ks, db ? extra : nullptr);
    if (db) launch_split_reduce(st, extra, p.nsplit, out, db);
  } else {
    if (db) launch_rowsum(st, Delta, out, B, extra, db);
    switch (p.bm) {
      case 96: launch_gemm<96, 128, 0, 0, EPI_RAW>(st, Delta, out, Hprev, in, part, out, out, in, B, p.nsplit, p.ks, nullptr, 0); break;
      case 128: launch_gemm<128, 128, 0, 0, EPI_RAW>(st, Delta, out, Hprev, in, part, out, out, in, B, p.nsplit, p.ks, nullptr, 0); break;
      default: launch_gemm<64, 128, 0, 0, EPI_RAW>(st, Delta, out, Hprev, in, part, out, out, in, B, p.nsplit, p.ks, nullptr, 0); break;
    }
  }
  launch_split_reduce(st, part, p.nsplit, (int64_t)out * in, dW);
}

// K3 for wide subspaces (M > 32), on the matrix cores:  P[N x M] = A[N x K] * V[K x M]  with  A(m = row, k) = A[m + ldA*k]
// (row-fast) and B(k, n) = V[n + Mpad*k] (row-fast), no split.  The VALU kernel of kernels_gram.hip holds M accumulators
// per row in registers, needs one pass over A per 32 columns and runs at 2 waves per SIMD: at cfg5 (K = 128, M = 64,
// 6.4 M rows per GPU) it took 25 ms; this is the same 105 GFLOP as MFMA work on one pass over A.
void launch_project_mfma(hipStream_t st, const double* A, int64_t ldA, int64_t N, int64_t K, const double* V, int32_t M,
                         int32_t Mpad, double* P, int64_t ldP) {
  const int64_t ks = (K + 15) / 16 * 16;
  for (int32_t m0 = 0; m0 < M; m0 += 64) {   // 64-column panels of P (one panel up to M = 64)
    const int32_t mc = M - m0 < 64 ? M - m0 : 64;
    launch_gemm<128, 64, 0, 0, EPI_RAW>(st, A, ldA, V + m0, Mpad, P + (int64_t)m0 * ldP, ldP, (int)std::min<int64_t>(N, 0x7fffffff),
                                        mc, K, 1, ks, nullptr, 0);
  }
}

// ------------------------------------------------------------------------------------------------ small kernels
// dst[e] = sum_split part[split][e], fixed order (bit-reproducible)
__global__ __launch_bounds__(256) void split_reduce_kernel(const double* __restrict__ part, int nsplit, int64_t elems,
                                                           double* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < elems; e += stride) {
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(int64_t)sp * elems + e];
    dst[e] = s;
  }
}
void launch_split_reduce(hipStream_t st, const double* part, int nsplit, int64_t elems, double* dst) {
  int64_t blocks = (elems + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, nsplit, elems, dst);
}

// Delta_L[e] = scale * (Y[e] - Yhat[e]) * act_L'(Yhat[e])      (scale = 1/sigma^2 for lp, -2/d for the mse loss)
__global__ __launch_bounds__(256) void delta_out_kernel(const double* __restrict__ Y, const double* __restrict__ Yhat,
                                                        int64_t d, double scale, int act, double* __restrict__ delta) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < d; e += stride) {
    const double yh = Yhat[e];
    delta[e] = scale * (Y[e] - yh) * dact_full(yh, act);
  }
}
void launch_delta_out(hipStream_t st, const double* Y, const double* Yhat, int64_t d, double scale, int act, double* delta) {
  int64_t blocks = (d + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(delta_out_kernel, dim3((unsigned)blocks), dim3(256), 0, st, Y, Yhat, d, scale, act, delta);
}

// db[i] = sum_b Delta[i + out*b]: stage 1 gives one partial per (column chunk, i); stage 2 sums chunks in order
__global__ __launch_bounds__(256) void rowsum_partial_kernel(const double* __restrict__ D, int out, int64_t B,
                                                             double* __restrict__ part) {
  // block = (32 rows) x (8 column lanes); grid.x = row groups, grid.y = column chunk
  __shared__ double red[8][33];
  const int il = threadIdx.x & 31, bl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il;
  const int64_t per = (B + RS_CHUNKS - 1) / RS_CHUNKS;
  const int64_t b0 = (int64_t)blockIdx.y * per;
  int64_t b1 = b0 + per;
  if (b1 > B) b1 = B;
  double s = 0.0;
  if (i < out)
    for (int64_t b = b0 + bl; b < b1; b += 8) s += D[i + (int64_t)out * b];
  red[bl][il] = s;
  __syncthreads();
  if (bl == 0 && i < out) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][il];
    part[(int64_t)blockIdx.y * out + i] = t;
  }
}
__global__ __launch_bounds__(256) void rowsum_final_kernel(const double* __restrict__ part, int out, double* __restrict__ db) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= out) return;
  double s = 0.0;
  for (int ch = 0; ch < RS_CHUNKS; ++ch) s += part[(int64_t)ch * out + i];
  db[i] = s;
}
// the same for a handful of rows (regression heads: out = 1): the 256 threads of a block walk the COLUMNS of one row's chunk
// (rowsum_partial_kernel keeps 8 threads per row busy: 85 us for 1 x 100 000, latency-bound), fixed-order tree in LDS
__global__ __launch_bounds__(256) void rowsum_narrow_kernel(const double* __restrict__ D, int out, int64_t B,
                                                            double* __restrict__ part) {
  __shared__ double red[256];
  const int i = blockIdx.x;
  const int64_t per = (B + RS_CHUNKS - 1) / RS_CHUNKS;
  const int64_t b0 = (int64_t)blockIdx.y * per;
  int64_t b1 = b0 + per;
  if (b1 > B) b1 = B;
  double s = 0.0;
  for (int64_t b = b0 + threadIdx.x; b < b1; b += 256) s += D[i + (int64_t)out * b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(int64_t)blockIdx.y * out + i] = red[0];
}
void launch_rowsum(hipStream_t st, const double* D, int32_t out, int64_t B, double* part /* RS_CHUNKS*out */, double* db) {
  if (out <= 8) {
    hipLaunchKernelGGL(rowsum_narrow_kernel, dim3(out, RS_CHUNKS), dim3(256), 0, st, D, (int)out, B, part);
    hipLaunchKernelGGL(rowsum_final_kernel, dim3(1), dim3(256), 0, st, part, (int)out, db);
    return;
  }
  hipLaunchKernelGGL(rowsum_partial_kernel, dim3((out + 31) / 32, RS_CHUNKS), dim3(256), 0, st, D, (int)out, B, part);
  hipLaunchKernelGGL(rowsum_final_kernel, dim3((out + 255) / 256), dim3(256), 0, st, part, (int)out, db);
}
int rowsum_chunks() { return RS_CHUNKS; }

// Reverse sweep through a NARROW last layer (out_last <= 4: regression heads) in ONE pass over its input H (F x B):
//   Delta_prev[i,b] = (sum_o W[o,i] Delta[o,b]) * act_prev'(H[i,b])      (written, F x B)
//   dW[o,i]         = sum_b Delta[o,b] H[i,b]                            (chunk partials, summed in fixed order)
//   db_prev[i]      = sum_b Delta_prev[i,b]                              (chunk partials, summed in fixed order)
// The generic route runs three GEMM-shaped launches with a degenerate dimension plus a row-sum pass, each reading or
// writing the 768 MB activation again (cfg2: 0.95 ms); this kernel is bound by one read of H and one write of
// Delta_prev.  A thread owns E consecutive features (16-B accesses when F is even) and walks its chunk of columns;
// Delta[:, b] is block-uniform.
constexpr int TAILB_CHUNKS = 512;
template <int OL, int E>
__global__ __launch_bounds__(256) void tail_bwd_kernel(const double* __restrict__ W, const double* __restrict__ Delta,
                                                       const double* __restrict__ H, int F, int64_t B, int act_prev,
                                                       double* __restrict__ DeltaPrev, double* __restrict__ part,
                                                       int64_t cols) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * E;
  if (i >= F) return;
  const int64_t b0 = (int64_t)blockIdx.y * cols;
  int64_t b1 = b0 + cols;
  if (b1 > B) b1 = B;
  double w[OL][E], dw[OL][E], db[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    db[e] = 0.0;
#pragma unroll
    for (int o = 0; o < OL; ++o) {
      w[o][e] = (i + e < F) ? W[o + (int64_t)OL * (i + e)] : 0.0;
      dw[o][e] = 0.0;
    }
  }
#pragma unroll 4
  for (int64_t b = b0; b < b1; ++b) {
    double h[E];
    if constexpr (E == 2) {
      const double2 v = *reinterpret_cast<const double2*>(H + i + (int64_t)F * b);
      h[0] = v.x;
      h[1] = v.y;
    } else {
      h[0] = H[i + (int64_t)F * b];
    }
    double dl[OL];
#pragma unroll
    for (int o = 0; o < OL; ++o) dl[o] = Delta[o + (int64_t)OL * b];
    double d[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      double t = 0.0;
#pragma unroll
      for (int o = 0; o < OL; ++o) {
        t = fma(w[o][e], dl[o], t);
        dw[o][e] = fma(dl[o], h[e], dw[o][e]);
      }
      d[e] = t * dact_from_output(h[e], act_prev);
      db[e] += d[e];
    }
    if constexpr (E == 2)
      *reinterpret_cast<double2*>(DeltaPrev + i + (int64_t)F * b) = make_double2(d[0], d[1]);
    else
      DeltaPrev[i + (int64_t)F * b] = d[0];
  }
  double* pp = part + (int64_t)blockIdx.y * (OL + 1) * F;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (i + e >= F) continue;
#pragma unroll
    for (int o = 0; o < OL; ++o) pp[(int64_t)o * F + i + e] = dw[o][e];
    pp[(int64_t)OL * F + i + e] = db[e];
  }
}

// dW[o + OL*i] = sum_chunks part[c][o][i];  db_prev[i] = sum_chunks part[c][OL][i].  Block = 32 entries x 8 chunk lanes:
// lane j adds chunks j, j+8, ... in order, the eight lane sums are added in a fixed tree => deterministic.
__global__ __launch_bounds__(256) void tail_bwd_reduce_kernel(const double* __restrict__ part, int chunks, int OL, int F,
                                                              double* __restrict__ dW, double* __restrict__ dbprev) {
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
      dW[o + (int64_t)OL * i] = t;
    else
      dbprev[i] = t;
  }
}

size_t tail_bwd_part_elems(int32_t out_last, int32_t F) { return (size_t)TAILB_CHUNKS * (out_last + 1) * F; }

void launch_tail_bwd(hipStream_t st, const double* W, const double* Delta, const double* H, int32_t out_last, int32_t F,
                     int64_t B, int32_t act_prev, double* DeltaPrev, double* part, double* dW, double* dbprev) {
  int chunks = TAILB_CHUNKS;
  if (chunks > B) chunks = (int)B;
  const int64_t cols = (B + chunks - 1) / chunks;
  chunks = (int)((B + cols - 1) / cols);
  const bool vec = (F % 2 == 0) && ((reinterpret_cast<uintptr_t>(H) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(DeltaPrev) & 15u) == 0);
  const int per = 256 * (vec ? 2 : 1);
  const dim3 grid((F + per - 1) / per, chunks);
#define SI_TAILB(OLV)                                                                                                  \
  if (vec)                                                                                                             \
    hipLaunchKernelGGL((tail_bwd_kernel<OLV, 2>), grid, dim3(256), 0, st, W, Delta, H, (int)F, B, (int)act_prev,       \
                       DeltaPrev, part, cols);                                                                         \
  else                                                                                                                 \
    hipLaunchKernelGGL((tail_bwd_kernel<OLV, 1>), grid, dim3(256), 0, st, W, Delta, H, (int)F, B, (int)act_prev,       \
                       DeltaPrev, part, cols)
  switch (out_last) {
    case 1: SI_TAILB(1); break;
    case 2: SI_TAILB(2); break;
    case 3: SI_TAILB(3); break;
    default: SI_TAILB(4); break;
  }
#undef SI_TAILB
  const int total = (out_last + 1) * F;
  hipLaunchKernelGGL(tail_bwd_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, st, part, chunks, (int)out_last,
                     (int)F, dW, dbprev);
}

// grad_z[m] = sum_r P[r + ldP*m] * g[r]   (HBM-bound: P is read once, N*M*8 bytes)
// Four columns of P per sweep over a block's rows (g read once per four columns, one reduction round per four), fixed
// summation order: thread-strided partial sums, xor-free shuffle tree, the four wave sums in order.
constexpr int PTG_BLOCKS = 1024;
constexpr int PTG_MT = 4;
__global__ __launch_bounds__(256) void ptg_partial_kernel(const double* __restrict__ P, int64_t ldP, int64_t N, int M,
                                                          const double* __restrict__ g, double* __restrict__ part) {
  __shared__ double red[4][PTG_MT];
  const int64_t per = ((N + PTG_BLOCKS - 1) / PTG_BLOCKS + 1) / 2 * 2;
  const int64_t r0 = (int64_t)blockIdx.x * per;
  int64_t r1 = r0 + per;
  if (r1 > N) r1 = N;
  for (int m0 = 0; m0 < M; m0 += PTG_MT) {
    double s[PTG_MT];
#pragma unroll
    for (int j = 0; j < PTG_MT; ++j) s[j] = 0.0;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
      const double gv = g[r];
#pragma unroll
      for (int j = 0; j < PTG_MT; ++j)
        if (m0 + j < M) s[j] += P[r + ldP * (m0 + j)] * gv;
    }
#pragma unroll
    for (int j = 0; j < PTG_MT; ++j) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_down(s[j], off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int j = 0; j < PTG_MT; ++j) red[threadIdx.x >> 6][j] = s[j];
    }
    __syncthreads();
    if (threadIdx.x < PTG_MT && m0 + threadIdx.x < M)
      part[(int64_t)blockIdx.x * M + m0 + threadIdx.x] =
          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}
// one workgroup per m: 256 threads stride over the PTG_BLOCKS partials, fixed tree (a serial loop over 1024 dependent loads
// per thread took 58 us)
__global__ __launch_bounds__(256) void ptg_final_kernel(const double* __restrict__ part, int M, double* __restrict__ gz) {
  __shared__ double red[4];
  const int m = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < PTG_BLOCKS; b += 256) s += part[(int64_t)b * M + m];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) gz[m] = (red[0] + red[1]) + (red[2] + red[3]);
}
void launch_ptg(hipStream_t st, const double* P, int64_t ldP, int64_t N, int M, const double* g, double* part, double* gz) {
  hipLaunchKernelGGL(ptg_partial_kernel, dim3(PTG_BLOCKS), dim3(256), 0, st, P, ldP, N, M, g, part);
  hipLaunchKernelGGL(ptg_final_kernel, dim3(M), dim3(256), 0, st, part, M, gz);
}
int ptg_blocks() { return PTG_BLOCKS; }

}  // namespace si
