// K2 Gram matrix G = A'A for K <= 208 in ONE pass over A (round 2): the LDS-DMA fed, one-wave-per-SIMD kernel.
// (kernels_gram.hip keeps the 128-column panel kernels for wider A, the projection K3 and launch_gram, which dispatches
// here.)  This file is compiled three times (-DSI_GW_PART=0/1/2, like eig.cpp): the tile counts NT = 1..13 are separate
// template instantiations with fully unrolled slab loops, and one translation unit holding all of them takes minutes.
//
// Decomposition.  4-wave workgroups, ONE per CU, one wave per SIMD: the upper-triangular 16x16 tile pairs of G are cut
// into PS = 4 / KS contiguous chunks (row-major pair order) and the 8 k steps of a 32-row slab into KS residue classes;
// wave (kg, g) accumulates chunk g over the k steps s = kg (mod KS).  Every accumulator of the chunk lives in AGPRs (up
// to 28 tiles = 224 registers), a k step costs ONE ds_read_b64 per column tile the chunk touches for CNT MFMAs.  (The
// 8-wave kernel of round 1 re-read every operand in each of its 8 waves, lost a third of its LDS cycles to bank
// conflicts and 12 % to the uneven deal of 28 pairs over 8 waves.)  KS = 4 up to K = 112 (all pairs in every wave:
// perfectly even), KS = 2 up to K = 160, KS = 1 (pairs only) up to K = 208: A is read once for every K <= 208.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "si_internal.h"

#ifndef SI_GW_KNOB   // compile-time knock-outs of tools/r05_gram_conflicts.sh; 0 in every shipped build
#define SI_GW_KNOB 0
#endif
#ifndef SI_GW_PART
#error "compile with -DSI_GW_PART=0|1|2"
#endif

namespace si {

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int GR = 32;       // slab rows
constexpr int GW_WAVES = 4;

__host__ __device__ constexpr int tri_a(int p, int nt) {
  int a = 0;
  while (p >= nt - a) {
    p -= nt - a;
    ++a;
  }
  return a;
}
__host__ __device__ constexpr int tri_b(int p, int nt) {
  int a = 0;
  while (p >= nt - a) {
    p -= nt - a;
    ++a;
  }
  return a + p;
}

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// The tile pairs are dealt to the PS chunks of a workgroup by COUNT.  (Dealing them by issue cost -- a pair of a ragged last
// column at 0.3 of a full one -- was built and measured: 13 / 15 pairs per chunk at K = 100 took 0.38 ms against 0.30 ms
// for 14 / 14, and the same happened at K = 104 with 16 / 12: the time of this kernel follows the LARGEST PAIR COUNT of a
// wave, not the MFMA issue cycles of its chunk; profiles/r03_gram_ragged_ab.log.)
__host__ __device__ constexpr int gw_bound(int g, int nt, int ps) {   // first pair of chunk g; gw_bound(ps) = P
  const int P = nt * (nt + 1) / 2;
  return g <= 0 ? 0 : g >= ps ? P : g * P / ps;
}
__host__ __device__ constexpr int gw_cmax(int nt, int ps) {
  int m = 0;
  for (int g = 0; g < ps; ++g) {
    const int c = gw_bound(g + 1, nt, ps) - gw_bound(g, nt, ps);
    m = c > m ? c : m;
  }
  return m;
}

// NQ: width of the LAST column tile in 4-column blocks.  NQ = 4: a full 16-column tile (or one whose padding is not worth
// a special case); NQ = 1: a RAGGED last tile (K mod 16 in 1..4; NQ = 2 works too but measured slower than the plain
// kernel at K = 104 and is not instantiated) -- its pairs (a, NT-1) are multiplied with
// v_mfma_f64_4x4x4_4b_f64 instead of the 16x16x4 instruction: one instruction = four independent 4x4x4 blocks =
// 16 rows of tile a against ONE 4-column block of the last tile, at ~0.3 of the issue time of a 16x16x4 (measured:
// tools/mfma_f64_4x4_probe.hip, 39.9 against 132 cycles on the same clock).  At K = 100 the last column of tile pairs
// costs 7 x 0.3 instead of 7 issue slots: 23.1 instead of 28 per k step (measured gain of the whole kernel at cfg2: 0.261
// -> 0.240 ms, see gw_bound below for why not more).  Operand layout of the 4x4x4_4b instruction
// (probed, the guide has no table for it): A_b[i][k] in lane i + 4b + 16k, B_b[k][j] in lane j + 4b + 16k, D_b[i][j] in
// lane j + 4b + 16i -- so the ordinary fragment of tile a IS the A operand (block b = rows 4b..4b+3 of the tile), and the
// B operand is a 4-column block of the last tile repeated in all four blocks (its own LDS read, `fr`).
template <int NT, int KS, int G, int NQ = 4>
struct GWave {
  static constexpr bool RAG = NQ < 4;
  static constexpr int NQR = RAG ? NQ : 1;
  static constexpr int PS = GW_WAVES / KS;
  static constexpr int P = NT * (NT + 1) / 2;
  static constexpr int cmax() { return gw_cmax(NT, PS); }
  static __device__ __forceinline__ int bound_rt(int g) { return g <= 0 ? 0 : g >= PS ? P : g * P / PS; }
  static constexpr int LO = gw_bound(G, NT, PS), HI = gw_bound(G + 1, NT, PS), CNT = HI - LO;
  static constexpr unsigned mask() {
    unsigned m = 0;
    for (int p = LO; p < HI; ++p) m |= (1u << tri_a(p, NT)) | (1u << tri_b(p, NT));
    return m;
  }
  // FIRST: the very first k step of a workgroup starts the accumulators from the inline constant 0 (a separate
  // zero-initialisation would materialise 8 * CNT zeros in arch VGPRs before they move to the AGPRs the MFMAs use)
  // `hook(I)` runs right after MFMA I has been issued: the staging traffic of a slab is dealt out between the MFMAs, one
  // small piece at a time, so that it executes in the shadow of the matrix pipe instead of in one clump behind a group
  // does this wave's chunk hold a pair of the ragged last column?
  static constexpr bool has_rag() {
    if (!RAG) return false;
    for (int p = LO; p < HI; ++p)
      if (tri_b(p, NT) == NT - 1) return true;
    return false;
  }
  template <int I, bool FIRST, class HOOK>
  static __device__ __forceinline__ void mfma(const double (&f)[NT], const double (&fr)[NQR], d4 (&acc)[CNT > 0 ? CNT : 1], HOOK&& hook) {
    if constexpr (I < CNT) {
      constexpr int a = tri_a(LO + I, NT), b = tri_b(LO + I, NT);
      if constexpr (RAG && b == NT - 1) {
        // ragged column: component jb of the accumulator holds rows of tile a x column block jb of the last tile
        static_for<NQR>([&](auto JC) {
          constexpr int jb = decltype(JC)::value;
          if constexpr (FIRST)
            acc[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[a], fr[jb], 0.0, 0, 0, 0);
          else
            acc[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(f[a], fr[jb], acc[I][jb], 0, 0, 0);
        });
      } else if constexpr (FIRST) {
        acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[a], f[b], (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
      } else {
        acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[a], f[b], acc[I], 0, 0, 0);
      }
      // The hand placement (fragment reads behind MFMA 1, LDS-DMA pieces behind the following ones) must survive the
      // machine scheduler: with the two MFMA shapes of a ragged chunk it regrouped the MFMAs, hoisted every LDS-DMA piece
      // in front of them and put an s_waitcnt lgkmcnt(0) between the fragment reads and their group (measured: 0.30 -> 0.44
      // ms at K = 104).  A scheduling barrier after every MFMA and after its hook keeps the source order.
      if constexpr (RAG) __builtin_amdgcn_sched_barrier(0);
      hook(std::integral_constant<int, I>{});
      if constexpr (RAG) __builtin_amdgcn_sched_barrier(0);
      mfma<I + 1, FIRST>(f, fr, acc, hook);
    }
  }
  // tile element (row + 16 * column) that component `comp` of pair I's accumulator holds in this lane
  template <int I>
  static __device__ __forceinline__ int tile_elem(int lane, int comp) {
    constexpr int b = tri_b(LO + I, NT);
    if constexpr (RAG && b == NT - 1)
      return (4 * ((lane >> 2) & 3) + (lane >> 4)) + 16 * (4 * comp + (lane & 3));   // D_b[i][j]: lane j + 4b + 16i
    else
      return ((lane >> 4) + 4 * comp) + 16 * (lane & 15);
  }
  template <int I>
  static constexpr int comps() {
    return (RAG && tri_b(LO + I, NT) == NT - 1) ? NQR : 4;
  }
};

// ------------------------------------------------------------------------------------------------
// The same wave decomposition fed by LDS-DMA (global_load_lds_dwordx4) instead of register staging: the slabs of A go
// straight from HBM into a ring of NB LDS buffers, NB - 1 slabs ahead of the one being multiplied, retired by a COUNTED
// s_waitcnt vmcnt((NB-2)*NT) + raw s_barrier (a __syncthreads() would drain the ring with vmcnt(0)).  Register staging
// gives a load one slab time to come back -- at K = 100 the kernel asks HBM for 4.4 TB/s and a slab is 1.5 us: loads are
// late, the matrix pipe waits (measured: memory side alone 0.13 ms, MFMA side alone 0.20 ms, together 0.27 ms) -- and a
// second register stage is defeated by the compiler's wait-count placement (loop-carried copies of the staged registers).
// One wave-instruction writes 1 KiB of LDS contiguously (wave-uniform base + lane*16): lanes 16u..16u+15 bring the 16
// row pairs of column u of a 4-column group, so the LDS image is [column][32 rows] with NO padding; the bank conflicts of
// that image (every column starts on the same bank) are removed by permuting WHICH row pair a lane fetches: slot j of
// column c holds row pair j ^ (c & 15), and the operand read of (column c, row 4s + q) looks in slot (2s + (q >> 1)) ^ c
// -- the 16 columns of a 32-lane half then cover the 16 slot positions, i.e. all 64 banks exactly once.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(3))) const double* lds_d_ptr;
// (inline asm inside the nested lambdas of the body does not survive the host-side pass: helpers)
template <int OFF>
__device__ __forceinline__ double gw_lds_read_b64(const double* p) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(uintptr_t)(lds_d_ptr)p), "n"(OFF));
  return v;
}
template <int NT>
__device__ __forceinline__ void gw_lds_wait(double (&f)[NT]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(f[t]));
}

template <int NT, int KS, int KG, int G, int NB, int NQ>
__device__ __forceinline__ void gram_glds_body(const double* __restrict__ A, int64_t ldA, int64_t N, int K,
                                               double* __restrict__ Gpart, double* sA) {
  using GW = GWave<NT, KS, G, NQ>;
  constexpr int NQR = GW::NQR;
  constexpr int NC = NT * 16;
  constexpr int BUF = NC * GR;            // doubles per LDS buffer (32 rows per column, unpadded)
  constexpr int H = (GR / 4) / KS;        // k steps of a slab owned by this wave
  constexpr int CNT = GW::CNT > 0 ? GW::CNT : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int wave = KG + KS * G;   // the ROLE of this wave (which hardware wave plays it: gram_glds_kernel)
  const int q = lane >> 4, c = lane & 15;
  // LDS-DMA piece p of a slab: the 4 columns 16p + 4*wave .. +3; lane -> (column u = lane >> 4, slot j = lane & 15)
  const double* src[NT];
#pragma unroll
  for (int p = 0; p < NT; ++p) {
    const int col = 16 * p + 4 * wave + (lane >> 4);
    const int rp = (lane & 15) ^ (col & 15);
    src[p] = A + (int64_t)(col < K ? col : K - 1) * ldA + 2 * rp;   // columns past K: finite garbage that only reaches G entries >= K
  }
  auto issue_one = [&](int64_t roff, double* dst, auto PC) {
    constexpr int p = decltype(PC)::value;
#if SI_GW_KNOB & 1   // (tools/r05_gram_conflicts.sh: no staging traffic at all)
    (void)roff; (void)dst;
#else
    __builtin_amdgcn_global_load_lds(src[p] + roff, (lds_void_ptr)(dst + 16 * p * GR), 16, 0, 0);
#endif
  };
  auto issue = [&](int64_t slab, int buf) {
    const int64_t roff = slab * GR;
    double* dst = sA + buf * BUF + (4 * wave) * GR;
    static_for<NT>([&](auto PC) { issue_one(roff, dst, PC); });
  };
  // operand element (column 16t + c, row 4s + q) of buffer 0, for the H owned k steps
  int fb[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const int s = KG + KS * i;
    fb[i] = c * GR + 2 * ((2 * s + (q >> 1)) ^ c) + (q & 1);
#if SI_GW_KNOB & 2   // operand reads of 64 consecutive doubles: conflict-free whatever the image (wrong numbers, timing only)
    fb[i] = lane + 64 * i;
#elif SI_GW_KNOB & 4   // operand reads of the UNSWIZZLED image: every column on the same banks (the positive control)
    fb[i] = c * GR + 2 * (2 * s + (q >> 1)) + (q & 1);
#endif
    asm volatile("" : "+v"(fb[i]));   // keep the reads of neighbouring steps apart (no ds_read2_b64)
  }
  // ragged last tile: B operand of the 4x4x4_4b instruction = column 16 (NT-1) + 4 jb + (c & 3), row 4s + q, the same in all
  // four blocks (lanes that differ only in bits 2..3 read one address: an LDS broadcast)
  int fbr[H][NQR];
  if constexpr (GW::has_rag()) {
#pragma unroll
    for (int i = 0; i < H; ++i)
#pragma unroll
      for (int jb = 0; jb < NQR; ++jb) {
        const int s = KG + KS * i, cj = 4 * jb + (c & 3);
        fbr[i][jb] = (16 * (NT - 1) + cj) * GR + 2 * ((2 * s + (q >> 1)) ^ cj) + (q & 1);
      }
  }
  const int64_t nslab = (N + GR - 1) / GR;
  const int64_t stride = gridDim.x;
  int64_t slab = blockIdx.x;
  double f0[NT], f1[NT];
  double r0[NQR], r1[NQR];
  double* out = Gpart + (int64_t)blockIdx.x * GW::P * 256;
  if (slab >= nslab) {  // (the launcher never starts more blocks than slabs; keep the partial defined anyway)
    for (int e = tid; e < GW::P * 256; e += 64 * GW_WAVES) out[e] = 0.0;
    return;
  }
  const int64_t last = nslab - 1;
  auto clamp = [&](int64_t sl) { return sl < last ? sl : last; };   // past the end: re-fetch the last slab, never used
  // prologue: the ring is filled -- the current slab plus NB - 1 slabs ahead of it
#pragma unroll
  for (int b = 0; b < NB; ++b) issue(clamp(slab + b * stride), b);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * NT) : "memory");
  __builtin_amdgcn_s_barrier();
  auto load_frags_i = [&](const double* buf, auto IC, double(&f)[NT], double(&r)[NQR]) {
    constexpr int ii = decltype(IC)::value;
    const int b = fb[ii];
    static_for<NT>([&](auto TC) {
      constexpr int T = decltype(TC)::value;
      constexpr unsigned M = GW::mask();
      // (with a ragged last tile the ordinary fragment of tile NT-1 is only the A operand of its own diagonal pair)
      constexpr bool need = ((M >> T) & 1u) && !(GW::RAG && T == NT - 1 && !(GW::LO <= GW::P - 1 && GW::P - 1 < GW::HI));
#if SI_GW_KNOB & 16   // (round 4's form, kept for tools/r05_gram_conflicts.sh: the compiler's own reads)
      if constexpr (need) f[T] = buf[b + T * 16 * GR];
#else
      // one ds_read_b64 per tile, by hand: left to itself the load/store optimiser pairs the reads of two tiles (same base, 4 KiB
      // apart) into ds_read2st64_b64, which is banked mod 32 in 16-lane groups -- 2-way conflicts on this image, half of all
      // LDS-array cycles (profiles/r05_gram_conflicts.txt; same bits of G, same time: the conflicts were free)
      if constexpr (need) f[T] = gw_lds_read_b64<T * 16 * GR * 8>(buf + b);
#endif
    });
    if constexpr (GW::has_rag()) {
#pragma unroll
      for (int jb = 0; jb < NQR; ++jb) r[jb] = buf[fbr[ii][jb]];
    }
  };
  load_frags_i(sA, std::integral_constant<int, 0>{}, f0, r0);
  d4 acc[CNT];
  int ring = 0;   // buffer of the current slab
  // With ONE wave per SIMD nothing hides an instruction that is not an MFMA: an LDS-DMA piece costs 60-190 cycles of issue
  // and the matrix pipe drains 64 cycles after the last MFMA it was given.  So every memory instruction of a slab is
  // dealt out BETWEEN MFMAs: the fragments of the next k step behind MFMA 1 of a group, the NT pieces of the slab NB ahead
  // behind MFMAs 2 .. NT+1 of the last group (measured before: 73 % MFMA-busy with the pieces in one clump per slab).
  auto one_slab = [&](auto FIRSTSLAB) {
    constexpr bool first_slab = decltype(FIRSTSLAB)::value;
    const double* cur = sA + ring * BUF;
    const int nxt = ring + 1 == NB ? 0 : ring + 1;
    const int64_t roff = clamp(slab + (int64_t)NB * stride) * GR;
    double* pdst = sA + ring * BUF + (4 * wave) * GR;
    static_for<H>([&](auto IC) {
      constexpr int i = decltype(IC)::value;
      double(&fc)[NT] = (i & 1) ? f1 : f0;
      double(&fn)[NT] = (i & 1) ? f0 : f1;
      double(&rc)[NQR] = (i & 1) ? r1 : r0;
      double(&rn)[NQR] = (i & 1) ? r0 : r1;
      if constexpr (i == H - 1) {
        // every LDS read of this buffer has been issued: retire them and this wave's pieces of the NEXT slab, meet the other
        // waves; afterwards this buffer belongs to the slab NB ahead
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NB - 2) * NT) : "memory");
        __builtin_amdgcn_s_barrier();
      }
      constexpr int FRAG_AT = GW::CNT > 1 ? 1 : 0;   // MFMA behind which the next fragments are fetched
      auto hook = [&](auto MC) {
        constexpr int I = decltype(MC)::value;
        if constexpr (I == FRAG_AT) {
          if constexpr (i + 1 < H)
            load_frags_i(cur, std::integral_constant<int, i + 1>{}, fn, rn);
          else
            load_frags_i(sA + nxt * BUF, std::integral_constant<int, 0>{}, fn, rn);
        }
        if constexpr (i == H - 1 && I >= FRAG_AT + 1 && I - (FRAG_AT + 1) < NT) {
          issue_one(roff, pdst, std::integral_constant<int, I - (FRAG_AT + 1)>{});
        }
      };
#if !(SI_GW_KNOB & 16)
      // the compiler does not count hand-written LDS reads: wait for this group's fragments (issued a whole group ago) and make
      // every MFMA operand depend on the wait
      gw_lds_wait<NT>(fc);
#endif
      __builtin_amdgcn_s_setprio(1);
      if constexpr (first_slab && i == 0)
        GW::template mfma<0, true>(fc, rc, acc, hook);
      else
        GW::template mfma<0, false>(fc, rc, acc, hook);
      __builtin_amdgcn_s_setprio(0);
      if constexpr (GW::CNT == 0) {   // a wave without tile pairs still stages its share
        if constexpr (i + 1 < H)
          load_frags_i(cur, std::integral_constant<int, i + 1>{}, fn, rn);
        else
          load_frags_i(sA + nxt * BUF, std::integral_constant<int, 0>{}, fn, rn);
      }
      if constexpr (i == H - 1) {     // pieces that found no MFMA to hide behind (small chunks)
        constexpr int done = GW::CNT - (FRAG_AT + 1) > 0 ? (GW::CNT - (FRAG_AT + 1) < NT ? GW::CNT - (FRAG_AT + 1) : NT) : 0;
        static_for<NT - done>([&](auto PC) { issue_one(roff, pdst, std::integral_constant<int, done + decltype(PC)::value>{}); });
      }
    });
    if constexpr ((H & 1) == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t) f0[t] = f1[t];
#pragma unroll
      for (int t = 0; t < NQR; ++t) r0[t] = r1[t];
    }
    ring = nxt;
    slab += stride;
  };
  one_slab(std::true_type{});
  while (slab < nslab) one_slab(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail fetches must land before the buffers are reused
  // ---- partial sums of the KS waves of a chunk: through LDS, R tiles per round, fixed order ----
  // (a pair of the ragged column fills only the tile elements of its valid 4-column blocks; the others keep whatever
  // finite values were there and land in G entries >= K, which the second reduction stage never writes)
  if constexpr (KS == 1) {
    static_for<GW::CNT>([&](auto IC) {
      constexpr int i = decltype(IC)::value;
#pragma unroll
      for (int r = 0; r < GW::template comps<i>(); ++r) out[(GW::LO + i) * 256 + GW::template tile_elem<i>(lane, r)] = acc[i][r];
    });
  } else {
    constexpr int PS = GW::PS;
    constexpr int CMAX = GW::cmax();
    constexpr int TP = 272;   // a 16 x 16 tile at pitch 17: the 16 lanes of a ds_write_b64 group (one column each) on 16 banks
    constexpr int R = (NB * BUF) / (TP * GW_WAVES) > 0 ? (NB * BUF) / (TP * GW_WAVES) : 1;
    constexpr int ROUNDS = (CMAX + R - 1) / R;
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int i = rd * R + j;
        if (i < GW::CNT) {
          static_for<GW::CNT>([&](auto IC) {   // the pair index selects the lane map at compile time (empty chunks: nothing)
            constexpr int ii = decltype(IC)::value;
            if (ii == i) {
#pragma unroll
              for (int r = 0; r < GW::template comps<ii>(); ++r) {
                const int te = GW::template tile_elem<ii>(lane, r);
                sA[(wave * R + j) * TP + (te & 15) + 17 * (te >> 4)] = acc[ii][r];
              }
            }
          });
        }
      }
      __syncthreads();
      for (int e = tid; e < PS * R * 256; e += 64 * GW_WAVES) {
        const int g2 = e / (R * 256), rem = e - g2 * (R * 256);
        const int j = rem >> 8, el = rem & 255;
        const int lo2 = GW::bound_rt(g2), cnt2 = GW::bound_rt(g2 + 1) - lo2;
        const int i = rd * R + j;
        if (i < cnt2) {
          double sum = 0.0;
#pragma unroll
          for (int kg2 = 0; kg2 < KS; ++kg2) sum += sA[((g2 * KS + kg2) * R + j) * TP + (el & 15) + 17 * (el >> 4)];
          out[(lo2 + i) * 256 + el] = sum;
        }
      }
      __syncthreads();
    }
  }
}

template <int NT, int KS, int NB, int OCC, int NQ>
__global__ __launch_bounds__(64 * GW_WAVES, OCC) void gram_glds_kernel(const double* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                                     double* __restrict__ Gpart) {
  extern __shared__ double sA[];  // [NB][NT*16][32]
  // (Reversing the roles of the second workgroup of a CU, so that every SIMD carries one heavy and one light chunk, was
  // measured: no difference.)
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: gram_glds_body<NT, KS, 0 % KS, 0 / KS, NB, NQ>(A, ldA, N, K, Gpart, sA); break;
    case 1: gram_glds_body<NT, KS, 1 % KS, 1 / KS, NB, NQ>(A, ldA, N, K, Gpart, sA); break;
    case 2: gram_glds_body<NT, KS, 2 % KS, 2 / KS, NB, NQ>(A, ldA, N, K, Gpart, sA); break;
    default: gram_glds_body<NT, KS, 3 % KS, 3 / KS, NB, NQ>(A, ldA, N, K, Gpart, sA); break;
  }
}


#ifdef SI_DEV_KNOBS   // development build only (python build.py --dev): the variants behind the measurements in DESIGN.md section 4
// ------------------------------------------------------------------------------------------------
// Wave-specialised variant: the 4 MFMA waves (one per SIMD) issue NOTHING but LDS reads and MFMAs; NP extra PRODUCER
// waves issue every LDS-DMA piece of the ring.  Why: a global_load_lds instruction holds the issuing wave for 60-190
// cycles, and with one wave per SIMD the matrix pipe drains behind it wherever the piece is placed in the MFMA stream
// (measured: 73-75 % MFMA-busy in every single-role structure).  A producer wave blocked in issue costs nothing -- the
// MFMA wave of its SIMD keeps the issue port.  Protocol per slab t (every wave executes exactly one s_barrier per slab):
//   consumer: ... k steps of slab t, all LDS reads of buffer t % NB issued and returned -> barrier -> first fragments of
//             slab t + 1 (buffer (t + 1) % NB, landed by the producers' promise) ...
//   producer: s_waitcnt vmcnt((NB - 2) * PP)  [slab t + 1 has landed; slabs t + 2 .. t + NB - 1 may still fly]
//             -> barrier -> issue its PP pieces of slab t + NB into buffer t % NB (free: every consumer passed the barrier).
// PP = 4 NT / NP pieces per producer and slab; (NB - 1) * PP <= 63 (the vmcnt field).
// ------------------------------------------------------------------------------------------------
#ifdef SI_GRAM_TS
__device__ unsigned long long g_gram_ts[8][1024];   // development: s_memrealtime at every slab barrier of 8 sampled workgroups
void gram_spec_dump_ts(unsigned long long* host) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gram_ts), sizeof(g_gram_ts)); }
#endif
template <int NT, int KS, int NB>
struct GSpec {
  static constexpr int NC = NT * 16;
  static constexpr int BUF = NC * GR;
  static constexpr int P = NT * (NT + 1) / 2;
  static constexpr int PS = GW_WAVES / KS;
  static constexpr int CMAX = (P + PS - 1) / PS;
  static constexpr int R = KS == 1 ? 1 : ((NB * BUF) / (256 * GW_WAVES) > 0 ? (NB * BUF) / (256 * GW_WAVES) : 1);
  static constexpr int ROUNDS = KS == 1 ? 0 : (CMAX + R - 1) / R;
};

template <int NT, int KS, int NB, int NP, int J>
__device__ __forceinline__ void gram_spec_producer(const double* __restrict__ A, int64_t ldA, int64_t N, int K, double* sA, int dbg) {
  using S = GSpec<NT, KS, NB>;
  constexpr int PIECES = 4 * NT;                      // 1 KiB pieces (4 columns x 32 rows) of a slab
  constexpr int PP = (PIECES - J + NP - 1) / NP;      // pieces J, J + NP, ... of every slab are this wave's
  constexpr int PPMAX = (PIECES + NP - 1) / NP;
  static_assert((NB - 1) * PPMAX <= 63, "vmcnt field");
  const int lane = threadIdx.x & 63;
  const double* src[PP];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int col = 4 * (J + NP * i) + (lane >> 4);
    const int rp = (lane & 15) ^ (col & 15);
    src[i] = A + (int64_t)(col < K ? col : K - 1) * ldA + 2 * rp;
  }
  const int64_t nslab = (N + GR - 1) / GR;
  const int64_t stride = gridDim.x;
  int64_t slab = blockIdx.x;
  if (slab >= nslab) return;   // (block-uniform; the consumers leave the same way before any barrier)
  const int64_t last = nslab - 1;
  auto clamp = [&](int64_t sl) { return sl < last ? sl : last; };
  auto issue = [&](int64_t sl, int buf) {
    if (dbg & 2) return;                       // development: no DMA at all (consumers multiply whatever LDS holds)
    if (dbg & 1) sl = sl & 7;                  // development: every workgroup re-reads 8 slabs (cache-resident source)
    const int64_t roff = sl * GR;
    double* dst = sA + buf * S::BUF;
    static_for<PP>([&](auto IC) {
      constexpr int i = decltype(IC)::value;
      __builtin_amdgcn_global_load_lds(src[i] + roff, (lds_void_ptr)(dst + 4 * (J + NP * i) * GR), 16, 0, 0);
    });
  };
#pragma unroll
  for (int b = 0; b < NB; ++b) issue(clamp(slab + b * stride), b);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * PP) : "memory");
  __builtin_amdgcn_s_barrier();
  int ring = 0;
  while (slab < nslab) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * PP) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(clamp(slab + (int64_t)NB * stride), ring);
    ring = ring + 1 == NB ? 0 : ring + 1;
    slab += stride;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail fetches must land before the buffers are reused
  if constexpr (KS > 1) {
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < S::ROUNDS; ++rd) {
      __syncthreads();
      __syncthreads();
    }
  }
}

template <int NT, int KS, int KG, int G, int NB>
__device__ __forceinline__ void gram_spec_consumer(int64_t N, double* __restrict__ Gpart, double* sA, int dbg) {
  using GW = GWave<NT, KS, G>;
  using S = GSpec<NT, KS, NB>;
  constexpr int BUF = S::BUF;
  constexpr int H = (GR / 4) / KS;
  constexpr int CNT = GW::CNT > 0 ? GW::CNT : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c = lane & 15;
  int fb[H];
#pragma unroll
  for (int i = 0; i < H; ++i) {
    const int s = KG + KS * i;
    fb[i] = c * GR + 2 * ((2 * s + (q >> 1)) ^ c) + (q & 1);
#if SI_GW_KNOB & 2   // operand reads of 64 consecutive doubles: conflict-free whatever the image (wrong numbers, timing only)
    fb[i] = lane + 64 * i;
#elif SI_GW_KNOB & 4   // operand reads of the UNSWIZZLED image: every column on the same banks (the positive control)
    fb[i] = c * GR + 2 * (2 * s + (q >> 1)) + (q & 1);
#endif
    asm volatile("" : "+v"(fb[i]));
  }
  const int64_t nslab = (N + GR - 1) / GR;
  const int64_t stride = gridDim.x;
  int64_t slab = blockIdx.x;
  double f0[NT], f1[NT];
  double* out = Gpart + (int64_t)blockIdx.x * GW::P * 256;
  if (slab >= nslab) {
    for (int e = tid; e < GW::P * 256; e += 64 * GW_WAVES) out[e] = 0.0;
    return;
  }
  __builtin_amdgcn_s_barrier();   // the first slab has landed
  auto load_frags = [&](const double* buf, int b, double(&f)[NT]) {
    static_for<NT>([&](auto TC) {
      constexpr int T = decltype(TC)::value;
      constexpr unsigned M = GW::mask();
      if constexpr ((M >> T) & 1u) f[T] = buf[b + T * 16 * GR];
    });
  };
  load_frags(sA, fb[0], f0);
  d4 acc[CNT];
  int ring = 0;
  auto one_slab = [&](auto FIRSTSLAB) {
    constexpr bool first_slab = decltype(FIRSTSLAB)::value;
    const double* cur = sA + ring * BUF;
    const int nxt = ring + 1 == NB ? 0 : ring + 1;
    static_for<H>([&](auto IC) {
      constexpr int i = decltype(IC)::value;
      double(&fc)[NT] = (i & 1) ? f1 : f0;
      double(&fn)[NT] = (i & 1) ? f0 : f1;
      if constexpr (i == H - 1) {
        // every LDS read of this buffer has been issued: retire them, meet the producers (next slab landed, this buffer free)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SI_GRAM_TS
        const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef SI_GRAM_TS
        if (KG == 0 && G == 0 && (blockIdx.x & 31) == 0 && lane == 0) {
          const int64_t it = (slab - blockIdx.x) / stride;
          if (it < 512) {
            g_gram_ts[blockIdx.x >> 5][2 * it] = t_in;
            g_gram_ts[blockIdx.x >> 5][2 * it + 1] = __builtin_amdgcn_s_memrealtime();
          }
        }
#endif
      }
      constexpr int FRAG_AT = GW::CNT > 1 ? 1 : 0;
      auto hook = [&](auto MC) {
        constexpr int I = decltype(MC)::value;
        if constexpr (I == FRAG_AT) {
          if constexpr (i + 1 < H)
            load_frags(cur, fb[i + 1], fn);
          else
            load_frags(sA + nxt * BUF, fb[0], fn);
        }
      };
      if (dbg & 4) {   // development: memory side alone (no MFMA, no fragment reads; the barriers stay)
        if constexpr (first_slab && i == 0) {
#pragma unroll
          for (int z = 0; z < CNT; ++z) acc[z] = (d4){0.0, 0.0, 0.0, 0.0};
        }
      } else {
        const double rdummy[1] = {0.0};
        if constexpr (first_slab && i == 0)
          GW::template mfma<0, true>(fc, rdummy, acc, hook);
        else
          GW::template mfma<0, false>(fc, rdummy, acc, hook);
        if constexpr (GW::CNT == 0) {
          if constexpr (i + 1 < H)
            load_frags(cur, fb[i + 1], fn);
          else
            load_frags(sA + nxt * BUF, fb[0], fn);
        }
      }
    });
    if constexpr ((H & 1) == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t) f0[t] = f1[t];
    }
    ring = nxt;
    slab += stride;
  };
  one_slab(std::true_type{});
  while (slab < nslab) one_slab(std::false_type{});
  if constexpr (KS == 1) {
#pragma unroll
    for (int i = 0; i < GW::CNT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(GW::LO + i) * 256 + (q + 4 * r) + 16 * c] = acc[i][r];
  } else {
    constexpr int PS = GW::PS;
    constexpr int R = S::R;
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < S::ROUNDS; ++rd) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int i = rd * R + j;
        if (i < GW::CNT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sA[(wave * R + j) * 256 + (q + 4 * r) + 16 * c] = acc[i < CNT ? i : 0][r];
        }
      }
      __syncthreads();
      for (int e = tid; e < PS * R * 256; e += 64 * GW_WAVES) {
        const int g2 = e / (R * 256), rem = e - g2 * (R * 256);
        const int j = rem >> 8, el = rem & 255;
        const int lo2 = GW::bound_rt(g2), cnt2 = GW::bound_rt(g2 + 1) - lo2;
        const int i = rd * R + j;
        if (i < cnt2) {
          double sum = 0.0;
#pragma unroll
          for (int kg2 = 0; kg2 < KS; ++kg2) sum += sA[((g2 * KS + kg2) * R + j) * 256 + el];
          out[(lo2 + i) * 256 + el] = sum;
        }
      }
      __syncthreads();
    }
  }
}

template <int NT, int KS, int NB, int NP>
__global__ __launch_bounds__(64 * (GW_WAVES + NP), 1) void gram_spec_kernel(const double* __restrict__ A, int64_t ldA, int64_t N, int K,
                                                                          double* __restrict__ Gpart, int dbg) {
  extern __shared__ double sA[];  // [NB][NT*16][32]
  static_assert(NP == 1 || NP == 2, "one or two producer waves");
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: gram_spec_consumer<NT, KS, 0 % KS, 0 / KS, NB>(N, Gpart, sA, dbg); break;
    case 1: gram_spec_consumer<NT, KS, 1 % KS, 1 / KS, NB>(N, Gpart, sA, dbg); break;
    case 2: gram_spec_consumer<NT, KS, 2 % KS, 2 / KS, NB>(N, Gpart, sA, dbg); break;
    case 3: gram_spec_consumer<NT, KS, 3 % KS, 3 / KS, NB>(N, Gpart, sA, dbg); break;
    case 4: gram_spec_producer<NT, KS, NB, NP, 0>(A, ldA, N, K, sA, dbg); break;
    default:
      if constexpr (NP == 2) gram_spec_producer<NT, KS, NB, NP, 1>(A, ldA, N, K, sA, dbg);
      break;
  }
}

static bool gram_spec_enabled() {
  static const bool v = [] {
    const char* e = getenv("SI_GRAM_SPEC");
    return e && e[0] == '1';
  }();
  return v;
}
template <int NT, int KS, int NB, int NP>
static void launch_spec(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks) {
  constexpr size_t lds = (size_t)NB * NT * 16 * GR * sizeof(double);
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(gram_spec_kernel<NT, KS, NB, NP>), lds);
  static const int dbg = getenv("SI_GRAM_SPEC_DBG") ? atoi(getenv("SI_GRAM_SPEC_DBG")) : 0;
  hipLaunchKernelGGL((gram_spec_kernel<NT, KS, NB, NP>), dim3(nblocks), dim3(64 * (GW_WAVES + NP)), lds, st, A, ldA, N, K, tiles, dbg);
}

#endif  // SI_DEV_KNOBS

// Up to K = 144 two workgroups per CU (two ring buffers each; 8 * pairs-per-wave + ~90 registers <= 256) are the shipped
// configuration: 0-6 % faster than one workgroup with a 4-deep ring (K = 100: 0.275 ms either way; K = 128: 2.22 against
// 2.33 ms at N = 6.4 M; K = 144: 0.411 against 0.437 ms).  (Development build: SI_GRAM_OCC1=1 selects the other one.)
#ifdef SI_DEV_KNOBS
static bool gram_two_per_cu() {
  static const bool v = [] {
    const char* e = getenv("SI_GRAM_OCC1");
    return !(e && e[0] == '1');
  }();
  return v;
}
#else
static constexpr bool gram_two_per_cu() { return true; }
#endif

// KS / KS2: k-step split with one / two workgroups per CU (two per CU needs 8 * pairs-per-wave + ~90 registers <= 256)
template <int NT, int KS, int KS2, int NQ>
static void launch_nt_q(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks) {
  if constexpr (KS2 > 0) {
    if (gram_two_per_cu()) {
      constexpr int NB = 2;
      constexpr size_t lds = (size_t)NB * NT * 16 * GR * sizeof(double);
      static LdsOptIn optin;
      optin.ensure(reinterpret_cast<const void*>(gram_glds_kernel<NT, KS2, NB, 2, NQ>), lds);
      hipLaunchKernelGGL((gram_glds_kernel<NT, KS2, NB, 2, NQ>), dim3(nblocks), dim3(64 * GW_WAVES), lds, st, A, ldA, N, K, tiles);
      return;
    }
  }
  // LDS-DMA ring: as many buffers as 150 KB hold, at most 4
  constexpr int NB = (150 * 1024) / (NT * 16 * GR * 8) >= 4 ? 4 : (150 * 1024) / (NT * 16 * GR * 8) >= 3 ? 3 : 2;
  constexpr size_t lds = (size_t)NB * NT * 16 * GR * sizeof(double);
  static LdsOptIn optin;
  optin.ensure(reinterpret_cast<const void*>(gram_glds_kernel<NT, KS, NB, 1, NQ>), lds);
  hipLaunchKernelGGL((gram_glds_kernel<NT, KS, NB, 1, NQ>), dim3(nblocks), dim3(64 * GW_WAVES), lds, st, A, ldA, N, K, tiles);
}

// the last column tile holds K - 16 (NT - 1) columns: 1..4 -> the ragged variant (one 4-column block on the 4x4x4_4b
// instruction), else the plain kernel
template <int NT, int KS, int KS2>
static void launch_nt(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks, int num_cu) {
  (void)num_cu;
  if (K - 16 * (NT - 1) <= 4)
    launch_nt_q<NT, KS, KS2, 1>(st, A, ldA, N, K, tiles, nblocks);
  else
    launch_nt_q<NT, KS, KS2, 4>(st, A, ldA, N, K, tiles, nblocks);
}

// workgroups per CU the launcher of this tile count will use (the caller sizes the grid and the partial-tile workspace)
#if SI_GW_PART == 0
int gram_wave_blocks_per_cu(int nt) {
#ifdef SI_DEV_KNOBS
  if (gram_spec_enabled() && (nt == 7 || nt == 8)) return 1;
#endif
  return (nt <= 9 && gram_two_per_cu()) ? 2 : 1;
}
#endif

// launches the partial-tile kernel for NT = ceil(K / 16) when this part holds it (tiles: nblocks x NT(NT+1)/2 x 256)
#if SI_GW_PART == 0
bool launch_gram_wave_part0(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks) {
  switch ((K + 15) / 16) {
    case 1: launch_nt<1, 4, 2>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 2: launch_nt<2, 4, 2>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 3: launch_nt<3, 4, 2>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 4: launch_nt<4, 4, 2>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 5: launch_nt<5, 4, 2>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 6: launch_nt<6, 4, 2>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 7:
#ifdef SI_DEV_KNOBS
      if (gram_spec_enabled()) {
        launch_spec<7, 2, 5, 2>(st, A, ldA, N, K, tiles, nblocks);
        return true;
      }
#endif
      launch_nt<7, 4, 2>(st, A, ldA, N, K, tiles, nblocks, 0);
      return true;
    case 8:
#ifdef SI_DEV_KNOBS
      if (gram_spec_enabled()) {
        launch_spec<8, 2, 4, 2>(st, A, ldA, N, K, tiles, nblocks);
        return true;
      }
#endif
      launch_nt<8, 2, 2>(st, A, ldA, N, K, tiles, nblocks, 0);
      return true;
    default: return false;
  }
}
#elif SI_GW_PART == 1
bool launch_gram_wave_part1(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks) {
  switch ((K + 15) / 16) {
    case 9: launch_nt<9, 2, 1>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 10: launch_nt<10, 2, 0>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 11: launch_nt<11, 1, 0>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    default: return false;
  }
}
#else
bool launch_gram_wave_part2(hipStream_t st, const double* A, int64_t ldA, int64_t N, int K, double* tiles, int nblocks) {
  switch ((K + 15) / 16) {
    case 12: launch_nt<12, 1, 0>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    case 13: launch_nt<13, 1, 0>(st, A, ldA, N, K, tiles, nblocks, 0); return true;
    default: return false;
  }
}
#endif

}  // namespace si
