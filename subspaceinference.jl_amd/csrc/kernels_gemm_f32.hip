// K5 in fp32 (compute_dtype = SI_F32, SURVEY section 0 Q6 / section 8(b),(d): "forward fp64 with fp32 as a measured option").
// Same operation as kernels_gemm.hip -- reference src/space_inference.jl:92-94, per Flux Dense layer
// H' = act.(W*H .+ b) with W (out x in) taken in place from the flat weight vector -- on the fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64 cycles per instruction per SIMD, 4096 flop: 157.3 TFLOP/s chip peak; exact fp32, one
// rounding per product, guide section 3).  Weights arrive ROUNDED ONCE from the fp64 W_swa + P z (K4 writes both), X is
// rounded once at set-up, activations are stored in fp32; the narrow head and the sum of squared errors stay in fp64.
//
// All matrices column-major (Julia):  W[i + out*k],  Hin[k + in*b],  Hout[i + out*b].
// One 32x32 MFMA tile has its ROWS on the batch index b and its COLUMNS on the feature index i:
//   A operand (32x2): lane l holds Hin[k = 2s + (l>>5)][b = l&31]
//   B operand (2x32): lane l holds   W[i = l&31][k = 2s + (l>>5)]
//   C/D: lane l, reg r holds D[b = (r&3) + 8*(r>>2) + 4*(l>>5)][i = l&31]
//
// Fast kernel (in % 16 == 0, out % 4 == 0, 16-byte aligned operands): the k tiles (16 deep) of both operands go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into TWO stages: per k tile
// s_waitcnt vmcnt(0), ONE raw s_barrier, issue of tile kt+1 into the stage tile kt-1 has just left, MFMAs of tile kt
// (the body also carries a ring of NB >= 3 stages with counted waits, and 32-deep tiles: measured, not shipped).  LDS images are LINEAR (the DMA writes
// wave-uniform base + lane*16 B):
//   sW[k][BM]   rows of BM floats;  the B fragment of k step (g, j), half h reads sW[8g + 4h + j][i0 + (l&31)]: 32
//               consecutive floats per half -> conflict-free ds_read_b32
//   sH[b][16]   64-byte rows, read by ds_read_b128: lane (b, h) takes the four k's 8g + 4h .. +3 at once, i.e. ONE read
//               feeds the four MFMA steps j = 0..3 of group g (step j multiplies k = 8g + j from half 0 with k = 8g + 4 + j
//               from half 1: a permutation of the k order inside the group, the same for both operands).  The 16-byte
//               slot of a row is XORed with (b >> 2) & 3 ON THE SOURCE SIDE of the DMA (guide rule 21: linear
//               destination + permuted source + the same involution on the read), which spreads the 16 lanes of every
//               ds_read_b128 lane group over all 16 slots of the 256-byte bank row.
// Rows past `out` / `B` are CLAMPED to a valid row in the DMA source (they only feed outputs that are never stored).
// Everything else runs through dense_f32_generic_kernel (register staging, zero fill, any shape).
// Roofline: MFMA fp32 (157.3 TFLOP/s); algorithmic flops = 2*out*in*B per layer.
#include <type_traits>

#include "kernels_gemm.h"

namespace si {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_ptr_f;

__device__ __forceinline__ float apply_act_f32(float v, int act) {
  switch (act) {
    case SI_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case SI_ACT_TANH: return tanhf(v);
    case SI_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

template <int N, int I = 0, class F>
__device__ __forceinline__ void f32_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    f32_static_for<N, I + 1>(f);
  }
}

// head of a regression chain folded into the epilogue (FUSE, as in kernels_gemm.hip): every wave reduces
// sum_i Wlast[o][i] * act(H[i][b]) over its own features IN FP64 and writes one partial per (feature slot, o, b)
template <int TM, int TN>
__device__ __forceinline__ void f32_fused_head(const f16v (&acc)[TM][TN], const int (&gi)[TM], int out, const float* __restrict__ Wlast,
                                               int out_last, double* __restrict__ part, int64_t part_ld, int64_t slot, int64_t bw0,
                                               int64_t B, int lane) {
  const int h = lane >> 5;
  for (int o = 0; o < out_last; ++o) {
    double wl[TM];
#pragma unroll
    for (int a = 0; a < TM; ++a) wl[a] = gi[a] < out ? (double)Wlast[o + (int64_t)out_last * gi[a]] : 0.0;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        double p = 0.0;
#pragma unroll
        for (int a = 0; a < TM; ++a) p = fma((double)acc[a][b][r], wl[a], p);
        // the 32 lanes of a half hold the 32 features of one tile row: butterfly sum inside the half
        p += __shfl_xor(p, 16, 32);
        p += __shfl_xor(p, 8, 32);
        p += __shfl_xor(p, 4, 32);
        p += __shfl_xor(p, 2, 32);
        p += __shfl_xor(p, 1, 32);
        const int64_t gb = bw0 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if ((lane & 31) == 0 && gb < B) part[(slot * out_last + o) * part_ld + gb] = p;
      }
    }
  }
}

// (MINW only keeps the body's instantiation 1:1 with its kernel's: two __global__ instantiations sharing one body with device
//  builtins inside lambdas do not survive the host-side pass)
template <int BM, int BN, int WM, int WN, int NB, int MINW, bool FUSE, int BK>
__device__ __forceinline__ void dense_f32_dma_body(const float* __restrict__ W, const float* __restrict__ bias,
                                                   const float* __restrict__ Hin, float* __restrict__ Hout, int out, int in,
                                                   int64_t B, int act, int nMt, int64_t nNt, const float* __restrict__ Wlast,
                                                   int out_last, double* __restrict__ part, const ChainBatch& cb,
                                                   const float* __restrict__ Dh, int dkind, float* smem) {
  constexpr int NWAVES = WM * WN;
  constexpr int KG = BK / 8;          // groups of 8 k values (one ds_read_b128 of H per half feeds four MFMA steps)
  constexpr int SPR = BK / 4;         // 16-byte slots per H row
  constexpr int TM = BM / WM / 32;   // feature tiles per wave
  constexpr int TN = BN / WN / 32;   // batch tiles per wave
  constexpr int NWI = BK * BM / 256; // LDS-DMA instructions (1 KiB each) per k tile: W part
  constexpr int NHI = BN * BK / 256; //                                               H part
  constexpr int NSLOT = (NWI + NHI + NWAVES - 1) / NWAVES;
  constexpr int STAGE = BK * BM + BN * BK;  // floats per ring stage
  static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0 && (BK * BM) % 256 == 0 && (BN * BK) % 256 == 0 && NB >= 2 && (BK == 16 || BK == 32), "tile shape");

  const int64_t bid = blockIdx.x;
  const int xcd = (int)(bid & 7);
  const int64_t jj = bid >> 3;
  const int mt = (int)(jj % nMt);
  const int64_t nt = (jj / nMt) * 8 + xcd;   // XCD-aware map: the nMt feature tiles of a batch panel share one L2
  if (nt >= nNt) return;                     // uniform per block: the whole workgroup leaves before any barrier

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int i0 = mt * BM;
  const int64_t b0 = nt * BN;
  const int h = lane >> 5, c = lane & 31;

  // ---- LDS-DMA plan of this wave: slot s is instruction id = wave + NWAVES*s of the tile's NWI + NHI instructions
  const float* src[NSLOT];
  int64_t adv[NSLOT];   // floats per k tile (wave-uniform)
  int dsto[NSLOT];      // LDS float offset inside a stage (wave-uniform)
  int nvalid = 0;
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int id = wave + NWAVES * s;
    if (id < NWI) {
      const int f = (id * 64 + lane) * 4;        // float offset inside sW: k = f / BM, i = f % BM
      const int k = f / BM, i = f % BM;
      int gi = i0 + i;
      if (gi > out - 4) gi = out - 4;            // clamped features feed outputs that are never stored
      src[s] = W + gi + (int64_t)out * k;
      adv[s] = (int64_t)out * BK;
      dsto[s] = id * 256;
      ++nvalid;
    } else if (id < NWI + NHI) {
      const int S = (id - NWI) * 64 + lane;      // 16-byte slot inside sH: row b = S / SPR, slot' = S % SPR
      const int b = S / SPR;
      const int slot = (S % SPR) ^ (BK == 16 ? (b >> 2) & 3 : (b >> 1) & 7); // source-side swizzle
      int64_t gb = b0 + b;
      if (gb > B - 1) gb = B - 1;
      src[s] = Hin + (int64_t)in * gb + 4 * slot;
      adv[s] = BK;
      dsto[s] = BK * BM + (id - NWI) * 256;
      ++nvalid;
    } else {
      src[s] = W;
      adv[s] = 0;
      dsto[s] = 0;
    }
  }
  nvalid = __builtin_amdgcn_readfirstlane(nvalid);
  const int nk = in / BK;
  auto issue = [&](int kt, int buf) {
    const int kc = kt < nk ? kt : nk - 1;   // past the end: re-fetch the last tile (keeps the vmcnt arithmetic uniform)
    float* dst = smem + buf * STAGE;
    f32_static_for<NSLOT>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      if (wave + NWAVES * s < NWI + NHI)
        __builtin_amdgcn_global_load_lds(src[s] + adv[s] * kc, (lds_void_ptr_f)(dst + dsto[s]), 16, 0, 0);
    });
  };

  f16v acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  // fragment addresses (floats inside a stage)
  const int wfo = wm * (BM / WM) + c;            // + (8g + 4h + j) * BM + 32 a
  int hfo[TN][KG];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int bl = wn * (BN / WN) + 32 * b + c;
#pragma unroll
    for (int g = 0; g < KG; ++g) hfo[b][g] = BK * BM + bl * BK + 4 * ((2 * g + h) ^ (BK == 16 ? (bl >> 2) & 3 : (bl >> 1) & 7));
  }

  auto compute = [&](const float* st) {
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      f4v hf[TN];
#pragma unroll
      for (int b = 0; b < TN; ++b) hf[b] = *reinterpret_cast<const f4v*>(st + hfo[b][g]);
      float wf[4][TM];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a) wf[j][a] = st[wfo + (8 * g + 4 * h + j) * BM + 32 * a];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(hf[b][j], wf[j][a], acc[a][b], 0, 0, 0);
    }
  };
  if constexpr (NB == 2) {
    // two stages (32-deep tiles: 2 x 40 KB per workgroup, two workgroups per CU): tile kt+1 is issued right behind the barrier
    // that retires tile kt-1 and has the whole compute time of tile kt to land
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
      compute(smem + (kt & 1) * STAGE);
    }
  } else {
    issue(0, 0);
    issue(1, 1);
    int ring = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt has landed once at most the DMAs of tile kt+1 (this wave's nvalid newest) are still in flight
      if (nvalid == NSLOT)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSLOT) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSLOT > 1 ? NSLOT - 1 : 0) : "memory");
      __builtin_amdgcn_s_barrier();   // everyone's pieces of tile kt are in; everyone is done reading tile kt-1
      const int nxt = ring + 2 >= NB ? ring + 2 - NB : ring + 2;
      issue(kt + 2, nxt);             // refills the stage of tile kt-1 (NB = 3) / an idle one (NB > 3)
      compute(smem + ring * STAGE);
      ring = ring + 1 == NB ? 0 : ring + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail fetches must land before the workgroup retires

#if defined(SI_F32_KNOB) && (SI_F32_KNOB & 1)   // tools/r05_f32_ceiling.sh only: no epilogue at all (the MFMAs stay observable)
  {
    float t = 0.0f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[a][b][r];
    if (t == 12345.678f) part[0] = 1.0;
    return;
  }
#endif
  // ---- epilogue: bias + activation (fp32), then either the store or the fused head
  const int iw0 = i0 + wm * (BM / WM);
  const int64_t bw0 = b0 + wn * (BN / WN);
  int gi[TM];
  float bv[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    gi[a] = iw0 + 32 * a + c;
    bv[a] = gi[a] < out ? bias[gi[a]] : 0.0f;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[a][b][r] + bv[a];
        acc[a][b][r] = act == SI_ACT_RELU ? (v > 0.0f ? v : 0.0f) : act == SI_ACT_IDENTITY ? v : apply_act_f32(v, act);
      }
  if constexpr (!FUSE) {
    // Store.  A lane holds D[b = (r&3) + 8*(r>>2) + 4h][i = c]: stored directly that is one 4-byte store per element in 128-byte
    // runs -- 16*TM*TN store instructions per wave, and on a shallow layer (cfg2's first: 8 k tiles) the store ISSUE is a
    // large part of the kernel.  Instead every wave transposes 8 batch rows at a time (the four registers 4g .. 4g+3 of both
    // halves) through its own slice of the now idle staging LDS and writes 16 bytes per lane, consecutive lanes on
    // consecutive features: a store instruction covers whole 384-byte rows of the wave's sub-tile (4x fewer instructions).
    constexpr int WI = BM / WM;                     // features per wave
    constexpr int NCH = 8 * WI / 4 / 64;            // 16-byte chunks per lane per 8-row block  (= TM)
    static_assert(NWAVES * 8 * WI <= NB * STAGE, "transposition scratch");
    __syncthreads();                                // every wave is done with the last k tile's fragments
    float* reg = smem + wave * (8 * WI);            // wave-private: no further workgroup barrier (LDS is in-order per wave)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) reg[(r + 4 * h) * WI + 32 * a + c] = acc[a][b][4 * g + r];
#pragma unroll
        for (int p = 0; p < NCH; ++p) {
          const int chunk = p * 64 + lane;
          const int row = chunk / (WI / 4), col4 = chunk % (WI / 4);
          f4v v = *reinterpret_cast<const f4v*>(reg + row * WI + 4 * col4);
          const int gf = iw0 + 4 * col4;
          const int64_t gb = bw0 + 32 * b + 8 * g + row;
          if (gf < out && gb < B) {   // out % 4 == 0: all four or none
            if (Dh != nullptr) {      // reverse sweep: this GEMM is W' Delta, and what is stored is (W' Delta) .* act'(H)
              const f4v hv = *reinterpret_cast<const f4v*>(Dh + gf + (int64_t)out * gb);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] *= dact_f32(hv[j], dkind);
            }
            *reinterpret_cast<f4v*>(Hout + gf + (int64_t)out * gb) = v;
          }
        }
      }
  } else if (Hout != nullptr) {   // (the gradient / training forward keeps this layer's output: plain stores)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gb = bw0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (gi[a] < out && gb < B) Hout[gi[a] + (int64_t)out * gb] = acc[a][b][r];
        }
  }
  if constexpr (FUSE)
    f32_fused_head<TM, TN>(acc, gi, out, Wlast, out_last, part, cb.part_ld ? cb.part_ld : B, (int64_t)mt * WM + wm, bw0, B, lane);
}

template <int BM, int BN, int WM, int WN, int NB, int MINW, bool FUSE, int BK = 16>
__global__ __launch_bounds__(64 * WM * WN, MINW) void dense_f32_dma_kernel(
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ Hin, float* __restrict__ Hout, int out,
    int in, int64_t B, int act, int nMt, int64_t nNt, const float* __restrict__ Wlast, int out_last, double* __restrict__ part,
    ChainBatch cb, const float* __restrict__ Dh, int dkind) {
  if (blockIdx.y != 0) {   // chain batching: every operand that differs per chain moves by its slot stride
    const int64_t ch = blockIdx.y;
    W += ch * cb.w;
    bias += ch * cb.w;
    Hin += ch * cb.hin;
    if (Hout != nullptr) Hout += ch * cb.hout;
    if (Dh != nullptr) Dh += ch * cb.hout;
    if constexpr (FUSE) {
      Wlast += ch * cb.w;
      part += ch * cb.part;
    }
  }
  extern __shared__ float smem_f32[];
  dense_f32_dma_body<BM, BN, WM, WN, NB, MINW, FUSE, BK>(W, bias, Hin, Hout, out, in, B, act, nMt, nNt, Wlast, out_last, part, cb, Dh,
                                                         dkind, smem_f32);
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: any shape and alignment (README toy: in = 10, out = 20, B = 100).  64 x 64 tiles, four waves with one
// 32x32 accumulator each, 8-deep k tiles staged through registers with bounds checks and zero fill.
// ------------------------------------------------------------------------------------------------
template <bool FUSE>
__global__ __launch_bounds__(256) void dense_f32_generic_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                                const float* __restrict__ Hin, float* __restrict__ Hout, int out,
                                                                int in, int64_t B, int act, int nMt, const float* __restrict__ Wlast,
                                                                int out_last, double* __restrict__ part, ChainBatch cb) {
  if (blockIdx.y != 0) {
    const int64_t ch = blockIdx.y;
    W += ch * cb.w;
    bias += ch * cb.w;
    Hin += ch * cb.hin;
    if (Hout != nullptr) Hout += ch * cb.hout;
    if constexpr (FUSE) {
      Wlast += ch * cb.w;
      part += ch * cb.part;
    }
  }
  constexpr int GM = 64, GN = 64, GK = 8;
  __shared__ float sW[GK][GM + 1];
  __shared__ float sH[GN][GK + 1];
  const int mt = (int)(blockIdx.x % nMt);
  const int64_t nt = blockIdx.x / nMt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int h = lane >> 5, c = lane & 31;
  const int i0 = mt * GM;
  const int64_t b0 = nt * GN;
  f16v acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int k0 = 0; k0 < in; k0 += GK) {
    for (int e = tid; e < GK * GM; e += 256) {
      const int k = e / GM, i = e % GM;
      sW[k][i] = (i0 + i < out && k0 + k < in) ? W[(i0 + i) + (int64_t)out * (k0 + k)] : 0.0f;
    }
    for (int e = tid; e < GN * GK; e += 256) {
      const int b = e / GK, k = e % GK;
      sH[b][k] = (b0 + b < B && k0 + k < in) ? Hin[(k0 + k) + (int64_t)in * (b0 + b)] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < GK / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sH[32 * wn + c][2 * s + h], sW[2 * s + h][32 * wm + c], acc, 0, 0, 0);
    __syncthreads();
  }
  const int gi1 = i0 + 32 * wm + c;
  const int64_t bw0 = b0 + 32 * wn;
  const float bv = gi1 < out ? bias[gi1] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = apply_act_f32(acc[r] + bv, act);
  if (!FUSE || Hout != nullptr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t gb = bw0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (gi1 < out && gb < B) Hout[gi1 + (int64_t)out * gb] = acc[r];
    }
  }
  if constexpr (FUSE) {
    const f16v accs[1][1] = {{acc}};
    const int gis[1] = {gi1};
    f32_fused_head<1, 1>(accs, gis, out, Wlast, out_last, part, cb.part_ld ? cb.part_ld : B, (int64_t)mt * 2 + wm, bw0, B, lane);
  }
}

struct FuseArgsF32 {
  const float* Wlast = nullptr;
  int out_last = 0;
  double* part = nullptr;
  ChainBatch cb;
  const float* dact_h = nullptr;   // !FUSE only: multiply the stored value by act'(dact_h[same element]) (the reverse sweep's dX)
  int dact_kind = SI_ACT_IDENTITY;
};

template <int BM, int BN, int WM, int WN, int NB, int MINW, bool FUSE, int BK = 16>
static void launch_f32_dma(hipStream_t st, const float* W, const float* bias, const float* Hin, float* Hout, int32_t out, int32_t in,
                           int64_t B, int32_t act, const FuseArgsF32& fa) {
  constexpr size_t lds = (size_t)NB * (BK * BM + BN * BK) * sizeof(float);
  const int nMt = (out + BM - 1) / BM;
  const int64_t nNt = (B + BN - 1) / BN;
  const int64_t groups = (nNt + 7) / 8;
  const int64_t grid = groups * nMt * 8;
  auto kern = dense_f32_dma_kernel<BM, BN, WM, WN, NB, MINW, FUSE, BK>;
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(kern), lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)fa.cb.n), dim3(64 * WM * WN), lds, st, W, bias, Hin, Hout, (int)out, (int)in, B,
                     (int)act, nMt, nNt, fa.Wlast, fa.out_last, fa.part, fa.cb, fa.dact_h, fa.dact_kind);
}

template <bool FUSE>
static void launch_f32_generic(hipStream_t st, const float* W, const float* bias, const float* Hin, float* Hout, int32_t out,
                               int32_t in, int64_t B, int32_t act, const FuseArgsF32& fa) {
  const int nMt = (out + 63) / 64;
  const int64_t nNt = (B + 63) / 64;
  hipLaunchKernelGGL((dense_f32_generic_kernel<FUSE>), dim3((unsigned)(nMt * nNt), (unsigned)fa.cb.n), dim3(256), 0, st, W, bias, Hin,
                     Hout, (int)out, (int)in, B, (int)act, nMt, fa.Wlast, fa.out_last, fa.part, fa.cb);
}

#ifndef SI_GEMM_F32_NO_DISPATCH
// the LDS-DMA kernel needs whole 16-deep k tiles and 16-byte pieces: in % 16 == 0, out % 4 == 0, aligned bases and slot strides
static bool f32_fast_shape(int32_t out, int32_t in) { return in >= 16 && in % 16 == 0 && out >= 4 && out % 4 == 0; }
static bool f32_fast_ok(const float* W, const float* Hin, const float* Hout, int32_t out, int32_t in, const ChainBatch& cb) {
  return f32_fast_shape(out, in) && (reinterpret_cast<uintptr_t>(W) & 15u) == 0 && (reinterpret_cast<uintptr_t>(Hin) & 15u) == 0 &&
         (reinterpret_cast<uintptr_t>(Hout) & 15u) == 0 && ((cb.w | cb.hin | cb.hout) & 3) == 0;
}
// feature tile of the fast kernel: the one that pads `out` least (960 = 5 x 192; 6656 = 52 x 128)
static int f32_pick_bm(int32_t out) {
  const int p192 = (out + 191) / 192 * 192, p128 = (out + 127) / 128 * 128;
  return p192 < p128 ? 192 : 128;
}

template <bool FUSE>
static void launch_f32_any(hipStream_t st, const float* W, const float* bias, const float* Hin, float* Hout, int32_t out, int32_t in,
                           int64_t B, int32_t act, const FuseArgsF32& fa) {
  if (!f32_fast_ok(W, Hin, Hout, out, in, fa.cb)) {
    launch_f32_generic<FUSE>(st, W, bias, Hin, Hout, out, in, B, act, fa);
    return;
  }
  // two stages of 16-deep tiles (measured against the three-stage ring with counted waits on one device, tools/f32_bk32.sh:
  // 1.45 against 1.49 ms on cfg2's 960 x 960 layer; 32-deep tiles in two stages: the same 1.45)
  if (f32_pick_bm(out) == 192)
    launch_f32_dma<192, 128, 2, 4, 2, 4, FUSE>(st, W, bias, Hin, Hout, out, in, B, act, fa);
  else
    launch_f32_dma<128, 128, 2, 4, 2, 4, FUSE>(st, W, bias, Hin, Hout, out, in, B, act, fa);
}

__global__ __launch_bounds__(256) void act_inplace_f32_kernel(float* __restrict__ H, int64_t n, int act) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) H[i] = (float)act_full((double)H[i], act);
}

void launch_dense_f32(hipStream_t st, const float* W, const float* bias, const float* Hin, float* Hout, int32_t out, int32_t in,
                      int64_t B, int32_t act, const ChainBatch& cb) {
  FuseArgsF32 fa;
  fa.cb = cb;
  if (act_is_extra(act)) {   // leakyrelu / elu / softplus / selu: identity in the GEMM, one elementwise pass behind it
    launch_f32_any<false>(st, W, bias, Hin, Hout, out, in, B, SI_ACT_IDENTITY, fa);
    for (int s = 0; s < cb.n; ++s) {
      const int64_t n = (int64_t)out * B;
      int64_t blocks = (n + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(act_inplace_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, Hout + (int64_t)s * cb.hout, n, (int)act);
    }
    return;
  }
  launch_f32_any<false>(st, W, bias, Hin, Hout, out, in, B, act, fa);
}

// The reverse sweep's dX: Dout = (Wt Delta) .* act'(Hprev) with the multiply in the GEMM's store (one read of Hprev instead of a
// read + write of the whole panel in a pass of its own).  Only the LDS-DMA kernel has that store: returns false when the shape
// went to the generic kernel and Dout = Wt Delta still needs its elementwise pass.
bool launch_dense_f32_dx(hipStream_t st, const float* Wt, const float* zero_bias, const float* Delta, float* Dout, int32_t out,
                         int32_t in, int64_t B, const float* Hprev, int32_t act_prev) {
  FuseArgsF32 fa;
  const bool fused = f32_fast_ok(Wt, Delta, Dout, out, in, fa.cb) && (reinterpret_cast<uintptr_t>(Hprev) & 15u) == 0;
  if (fused) {
    fa.dact_h = Hprev;
    fa.dact_kind = act_prev;
  }
  launch_f32_any<false>(st, Wt, zero_bias, Delta, Dout, out, in, B, SI_ACT_IDENTITY, fa);
  return fused;
}

// number of feature slots (partials per (o, b)) the fused kernel writes for a layer; `aligned` = what f32_fast_ok will see
int dense_f32_fused_slots(int32_t out, int32_t in, bool aligned) {
  if (aligned && f32_fast_shape(out, in)) {
    const int bm = f32_pick_bm(out);
    return (out + bm - 1) / bm * 2;
  }
  return (out + 63) / 64 * 2;
}

void launch_dense_f32_fused(hipStream_t st, const float* W, const float* bias, const float* Hin, int32_t out, int32_t in, int64_t B,
                            int32_t act, const float* Wlast, int32_t out_last, double* part, const ChainBatch& cb, float* Hkeep) {
  FuseArgsF32 fa;
  fa.Wlast = Wlast;
  fa.out_last = out_last;
  fa.part = part;
  fa.cb = cb;
  launch_f32_any<true>(st, W, bias, Hin, Hkeep, out, in, B, act, fa);
}

// ------------------------------------------------------------------------------------------------
// small fp32 helpers of the SI_F32 path
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void narrow_f64_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}
void launch_narrow_f32(hipStream_t st, const double* src, float* dst, int64_t n) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(narrow_f64_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n);
}

// sum (y - yhat)^2 with yhat in fp32, accumulated in fp64; optionally leaves yhat widened to fp64 (si_forward).
// Same fixed block count / fixed order as sse_partial_kernel: bit-reproducible.
__global__ __launch_bounds__(256) void sse_partial_f32_kernel(const float* __restrict__ yhat, const double* __restrict__ y, int64_t d,
                                                              double* __restrict__ part, int64_t yhat_stride,
                                                              double* __restrict__ yhat64, int64_t yhat64_stride) {
  __shared__ double red[4];
  double acc = 0.0;
  yhat += (int64_t)blockIdx.y * yhat_stride;
  part += (int64_t)blockIdx.y * gridDim.x;
  if (yhat64) yhat64 += (int64_t)blockIdx.y * yhat64_stride;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d; i += stride) {
    const double v = (double)yhat[i];
    if (yhat64) yhat64[i] = v;
    const double r = y[i] - v;
    acc += r * r;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
void launch_sse_f32(hipStream_t st, const float* yhat, const double* y, int64_t d, double* part, int nblocks, double* sse_out, int nch,
                    int64_t yhat_stride, double* yhat64, int64_t yhat64_stride, bool with_final) {
  hipLaunchKernelGGL(sse_partial_f32_kernel, dim3(nblocks, nch), dim3(256), 0, st, yhat, y, d, part, yhat_stride, yhat64, yhat64_stride);
  if (with_final) launch_sse_final(st, part, nblocks, sse_out, nch);
}
#endif

}  // namespace si
