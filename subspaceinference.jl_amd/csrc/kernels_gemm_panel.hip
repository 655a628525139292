// K5, the wide first layer of a regression MLP:  H = act.(W*X .+ b)  with a SHORT reduction (in <= 128) and a wide
// output (reference src/space_inference.jl:92-94 through Flux's Dense; cfg2: 128 -> 960 on 1e5 points, 768 MB of output).
//
// dense_f64_kernel (kernels_gemm.hip) runs this layer at 0.66 of the fp64 MFMA peak: its 96x128 tile has only 8 k tiles, and
// its 24 output stores per thread drain behind an idle matrix pipe -- loads and stores share ONE in-order counter (vmcnt) on gfx9,
// so whatever the wave does next with memory waits for the stores (k loop alone 0.39 ms, stores alone 0.17 ms,
// together 0.51 ms).  This kernel turns the loops inside out so that a store has several tiles' worth of MFMAs to drain:
//   * a workgroup (8 waves) owns a PANEL of 128 batch columns and a run of 16-row feature tiles; wave w owns 16 columns and
//     keeps their whole k extent in registers for the lifetime of the workgroup: the A operands of all in/4 MFMA steps
//     (lane (q, c) holds X[4s + q][b0 + 16w + c], 2 VGPRs per step, 64 VGPRs at in = 128);
//   * the W tiles (16 features x in, <= 16 KB) stream through an LDS ring of RD slots by LDS-DMA (global_load_lds_dwordx4: no
//     staging registers), RD - 1 tiles ahead of the MFMAs that read them; one s_barrier per tile;
//   * per tile a wave runs in/4 MFMAs on ONE accumulator (16 columns x 16 features) and issues its 4 stores; the wait in front
//     of the next tile is  s_waitcnt vmcnt((RD-1)*4 + (RD-2)*NDMA): the stores of the last RD - 1 tiles and the DMAs of the
//     next RD - 2 stay in flight.  A store is RD - 1 tiles old before anything waits for it.
// Every memory instruction is issued unconditionally (the vmcnt arithmetic counts instructions): lanes past an edge work on a
// CLAMPED column / feature, compute the very value its owner computes, and store it to the same address.
// Bit-identical to dense_f64_kernel: same instruction, same operand roles (A = activations, B = weights), k steps in the same
// order, same `acc + bias` then activation.
// Roofline: MFMA f64 (78.6 TFLOP/s); algorithmic flops 2*out*in*B; bytes in*B*8 read + out*B*8 written.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels_gemm.h"

namespace si {

typedef __attribute__((address_space(3))) void* lds_void_ptr_p;

// what the DMA fetches for the k rows past `in` of a reduction that is not a multiple of 16 (zeros, like the zero fill of
// dense_f64_kernel's ragged k tile: a clamped row of W would turn an Inf weight into 0 * Inf)
__device__ __attribute__((aligned(16))) double si_panel_zero[1024 + 16];   // (as long as the widest layer: a padded lane's address moves with the tile too)

#if defined(SI_PANEL_KNOB) && (SI_PANEL_KNOB & 8)   // harness only: per workgroup, shader cycles of the unit loop and its entry / loop start / loop end in 100 MHz ticks
__device__ long long si_panel_stamps[4 * 2048];
#endif

__device__ __forceinline__ double panel_act(double v, int act) {
  switch (act) {
    case SI_ACT_RELU: return v > 0.0 ? v : 0.0;
    case SI_ACT_TANH: return tanh(v);
    case SI_ACT_SIGMOID: return 1.0 / (1.0 + exp(-v));
    default: return v;
  }
}

template <int N, typename F>
__device__ __forceinline__ void panel_static_for(F&& f) {
  if constexpr (N > 0) {
    panel_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// The W fragments are read by hand: ds_read_b64 with the k step in the immediate offset, one chunk of four ahead of the MFMAs that
// use it, counted waits.  (Left to the compiler the reads are paired into ds_read2st64_b64 and each pair is waited for right in
// front of its two MFMAs -- and, worse, a compiler that SEES reads of LDS behind an LDS-DMA puts s_waitcnt vmcnt(0) in front of
// them: the DMA just issued and every store in flight.)
typedef __attribute__((address_space(3))) const double* panel_lds_cptr;
template <int OFF>
__device__ __forceinline__ double panel_lds_read(unsigned addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void panel_lds_wait(double (&a)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]));
}
template <int CH>
__device__ __forceinline__ void panel_read_chunk(unsigned fa, double (&a)[4]) {
  a[0] = panel_lds_read<512 * (4 * CH + 0)>(fa);
  a[1] = panel_lds_read<512 * (4 * CH + 1)>(fa);
  a[2] = panel_lds_read<512 * (4 * CH + 2)>(fa);
  a[3] = panel_lds_read<512 * (4 * CH + 3)>(fa);
}
template <int KT, int CH>
__device__ __forceinline__ void panel_chunks(unsigned fa, const double (&xf)[4 * KT], double (&cur)[4], double (&nxt)[4], d4& acc) {
  if constexpr (CH < KT) {
    if constexpr (CH + 1 < KT) {
      panel_read_chunk<CH + 1>(fa, nxt);
      panel_lds_wait<4>(cur);
    } else {
      panel_lds_wait<0>(cur);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[4 * CH + i], cur[i], acc, 0, 0, 0);
    panel_chunks<KT, CH + 1>(fa, xf, nxt, cur, acc);
  }
}

// KT = in / 16, RD = ring depth.  The launch is PERSISTENT: the npan * ntile (panel, feature tile) units are dealt out in equal
// contiguous runs to the G workgroups of the grid (two per CU), panel-major -- a run crosses a panel boundary a couple of times
// and reloads its X operands there; W tiles just keep streaming (tile index modulo ntile).  No tail round: every workgroup
// ends within one tile of every other.
template <int KT, int RD, bool RAG>   // RAG: in < 16 KT
__global__ __launch_bounds__(512, 4) void dense_f64_panel_kernel(const double* __restrict__ W, const double* __restrict__ bias,
                                                                  const double* __restrict__ Hin, double* __restrict__ Hout, int out,
                                                                  int in, int64_t B, int act, int ntile, int64_t units) {
  constexpr int IN = 16 * KT, KS = 4 * KT;          // the reduction padded to whole k tiles (in <= IN: rows past `in` are zeros), k steps of one tile
  constexpr int SLOT = IN * 16;                     // doubles per ring slot: [k][16 features]
  constexpr int NINST = 2 * KT;                     // DMA instructions per tile (1 KB each)
  constexpr int NDMA = (NINST + 7) / 8;             // per wave
  constexpr int NST = 4;                            // stores per tile and wave
  constexpr int NWAIT = (RD - 1) * NST + (RD - 2) * NDMA;
  static_assert(RD >= 2 && NWAIT <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ double ring[];                  // [RD][SLOT], then the bias as the tiles see it [ntile][16]

#if defined(SI_PANEL_KNOB) && (SI_PANEL_KNOB & 8)
  const long long sr_entry = __builtin_amdgcn_s_memrealtime();
#endif
  const int64_t u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  if (u0 >= u1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, c = lane & 15;
  int64_t pan = u0 / ntile;
  int t = (int)(u0 - pan * ntile);
  const int tail = 16 * ntile - out;                // the ragged last tile is fetched from features out - 16 .. out - 1 ...

  // the bias goes through LDS: a global load inside the loop would put the stores in front of its wait
  double* sbias = ring + RD * SLOT;
  for (int i = tid; i < 16 * ntile; i += 512) sbias[i] = bias[i >= 16 * (ntile - 1) ? i - tail : i];
  __syncthreads();

  // ---- DMA plan: instruction id covers k rows 8 id .. 8 id + 7 of a tile; lane -> (k = 8 id + lane / 8, feature pair lane % 8)
  const double* src[NDMA];
  int dsto[NDMA];
#pragma unroll
  for (int s = 0; s < NDMA; ++s) {
    int id = wave + 8 * s;
    if (id > NINST - 1) id = NINST - 1;             // a spare instruction repeats the last one (same data, same place)
    const int k = 8 * id + (lane >> 3);
    src[s] = RAG && k >= in ? si_panel_zero + 2 * (lane & 7) : W + (int64_t)out * k + 2 * (lane & 7);
    dsto[s] = id * 128;
  }
  auto issue = [&](int tt, int slot) {
    const int shift = tt == ntile - 1 ? 16 * tt - tail : 16 * tt;   // ... and then holds feature 16 tt + cc - tail in column cc
    panel_static_for<NDMA>([&](auto SC) {
      constexpr int s = decltype(SC)::value;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src[s] + shift), (lds_void_ptr_p)(ring + slot * SLOT + dsto[s]), 16, 0, 0);
    });
  };

  // ---- the wave's 16 columns, whole k extent, as MFMA A operands; the column bases of its stores (lane (q, c) holds
  //      D[column q + 4 r][feature c])
  double xf[KS];
  int so[4];                                        // element offsets from the panel's first output column
  double* Hp = Hout;
  auto load_panel = [&]() {
    int64_t gb = pan * 128 + 16 * wave + c;
    if (gb > B - 1) gb = B - 1;                     // (a column past the edge repeats column B - 1, value and address)
    if constexpr (RAG) {
      // (predicated loads off ONE base with the k step in the immediate offset: clamped per-element addresses cost the loop its
      //  registers -- the allocator then reloads the DMA pointers from scratch behind every barrier)
      const double* xp = Hin + (int64_t)in * gb + q;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        double xv = 0.0;
        if (4 * s + q < in) xv = xp[4 * s];
        xf[s] = xv;
      }
    } else {
      const double* xp = Hin + (int64_t)IN * gb + q;
#pragma unroll
      for (int s = 0; s < KS; ++s) xf[s] = xp[4 * s];
    }
    const int64_t left = B - 1 - pan * 128;         // last valid column of the panel
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int b = 16 * wave + q + 4 * r;
      if (b > left) b = (int)left;
      so[r] = b * out;
    }
    Hp = Hout + pan * 128 * out;
    // the operands are waited for HERE: a pending load that the compiler sees entering the tile loop puts s_waitcnt vmcnt(0) in
    // front of the first MFMA of every tile -- the DMA just issued and all stores in flight
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(xf[s]));
  };
  load_panel();
  int tpre = t;                                     // the tile the next DMA fetches
#pragma unroll
  for (int d = 0; d < RD - 1; ++d) {
    issue(tpre, d);
    tpre = tpre + 1 == ntile ? 0 : tpre + 1;
  }

#if defined(SI_PANEL_KNOB) && (SI_PANEL_KNOB & 8)
  const long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
  const unsigned frag0 = (unsigned)(uintptr_t)(panel_lds_cptr)(ring + q * 16 + c);   // LDS byte address of the lane's fragment column
  int slot = 0;
  for (int64_t u = u0; u < u1; ++u) {
    // the unit's tile has landed once only the newer traffic is in flight: 4 stores for each of the last RD - 1 units (fewer at
    // the start), NDMA loads for each of the next RD - 2.  (More behind it -- the X loads of a panel change -- only makes the
    // wait stricter.)
    const int64_t n = u - u0;
    const double bv = sbias[16 * t + c];            // (compiler's read: the wait below covers it)
    if (n >= RD - 1) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWAIT) : "memory");
    } else {
      panel_static_for<RD - 1>([&](auto NC) {
        constexpr int nc = decltype(NC)::value;
        if (n == nc) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RD - 2) * NDMA + NST * nc) : "memory");
      });
    }
#if !(defined(SI_PANEL_KNOB) && (SI_PANEL_KNOB & 4))
    __builtin_amdgcn_s_barrier();                   // everyone's pieces of this tile are in; everyone is done reading the one before
#endif
    const int nxt = slot == 0 ? RD - 1 : slot - 1;  // the slot the previous unit has just left
#if !(defined(SI_PANEL_KNOB) && (SI_PANEL_KNOB & 2))
    issue(tpre, nxt);                               // (past the end of the run: a fetch nobody reads keeps the arithmetic uniform)
#endif
    tpre = tpre + 1 == ntile ? 0 : tpre + 1;
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    {
      const unsigned fa = frag0 + slot * (SLOT * 8);
      double wa[4], wb[4];
      panel_read_chunk<0>(fa, wa);
      panel_chunks<KT, 0>(fa, xf, wa, wb, acc);
    }
    const int fi = t == ntile - 1 ? 16 * t + c - tail : 16 * t + c;
    double v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] + bv;
    if (act == SI_ACT_RELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0 ? v[r] : 0.0;
    } else if (act != SI_ACT_IDENTITY) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = panel_act(v[r], act);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#if defined(SI_PANEL_KNOB) && (SI_PANEL_KNOB & 1)
      if (v[r] == 1.2345e300) Hp[so[r] + fi] = v[r];
#else
      Hp[so[r] + fi] = v[r];
#endif
    }
    slot = slot + 1 == RD ? 0 : slot + 1;
    if (++t == ntile) {                             // next panel
      t = 0;
      ++pan;
      if (u + 1 < u1) load_panel();
    }
  }
#if defined(SI_PANEL_KNOB) && (SI_PANEL_KNOB & 8)
  if (tid == 0 && blockIdx.x < 2048) {
    si_panel_stamps[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0;
    si_panel_stamps[4 * blockIdx.x + 1] = sr_entry;
    si_panel_stamps[4 * blockIdx.x + 2] = sr0;
    si_panel_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail fetches must land before the workgroup retires
}

template <int KT>
static void launch_panel_inst(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout, int32_t out,
                              int32_t in, int64_t B, int32_t act, int grid) {
#ifdef SI_PANEL_RD
  constexpr int RD = SI_PANEL_RD;
#else
  constexpr int RD = 4;
#endif
  const int ntile = (out + 15) / 16;
  const int64_t npan = (B + 127) / 128, units = npan * ntile;
  if (grid > units) grid = (int)units;
  const size_t lds = ((size_t)RD * KT * 256 + 16 * (size_t)ntile) * sizeof(double);
  if (in == 16 * KT)
    hipLaunchKernelGGL((dense_f64_panel_kernel<KT, RD, false>), dim3((unsigned)grid), dim3(512), lds, st, W, bias, Hin, Hout, (int)out, (int)in, B, (int)act,
                       ntile, units);
  else
    hipLaunchKernelGGL((dense_f64_panel_kernel<KT, RD, true>), dim3((unsigned)grid), dim3(512), lds, st, W, bias, Hin, Hout, (int)out, (int)in, B, (int)act,
                       ntile, units);
}

// two workgroups per CU of the current device
static int panel_grid() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
  int g = cached[dev].load(std::memory_order_relaxed);
  if (g == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    g = 2 * cus;
    cached[dev].store(g, std::memory_order_relaxed);
  }
  return g;
}

// The class: a reduction of at most 128 (the X operands of a wave live in registers; padded with zeros to whole k tiles), 64 <= out <= 1024 and even with W
// 16-byte aligned (the DMA moves feature pairs), one of the four epilogue activations, and a batch that gives every workgroup of
// the persistent grid a run of at least 16 tiles (below that the prologue -- 15 us of a workgroup's life -- eats the gain).  (Development build: SI_PANEL=0 in the environment sends these layers to
// dense_f64_kernel -- the A/B runs of DESIGN section 10.8.)
bool dense_panel_applies(const double* W, int32_t out, int32_t in, int64_t B, int32_t act) {
#ifdef SI_DEV_KNOBS   // development build only (tests/test_capi_cpu.py: the shipped library reads no SI_* knobs)
  static const bool off = [] { const char* e = getenv("SI_PANEL"); return e && e[0] == '0'; }();
  if (off) return false;
#endif
  if (in < 1 || in > 128 || out < 64 || out > 1024 || out % 2 != 0 || act_is_extra(act) || B < 1) return false;
  if ((reinterpret_cast<uintptr_t>(W) & 15) != 0) return false;
  return (int64_t)((out + 15) / 16) * ((B + 127) / 128) >= 16 * (int64_t)panel_grid();
}

// true = launched (`grid` = workgroups, 0 = two per CU; the harness sweeps it and skips the size rule of dense_panel_applies).
bool launch_dense_f64_panel(hipStream_t st, const double* W, const double* bias, const double* Hin, double* Hout, int32_t out,
                            int32_t in, int64_t B, int32_t act, int grid) {
  if (in < 1 || in > 128 || out < 16 || out > 1024 || out % 2 != 0 || act_is_extra(act) || B < 1 || grid < 0) return false;
  if ((reinterpret_cast<uintptr_t>(W) & 15) != 0) return false;
  if (grid == 0) grid = panel_grid();
  switch ((in + 15) / 16) {
    case 1: launch_panel_inst<1>(st, W, bias, Hin, Hout, out, in, B, act, grid); break;
    case 2: launch_panel_inst<2>(st, W, bias, Hin, Hout, out, in, B, act, grid); break;
    case 3: launch_panel_inst<3>(st, W, bias, Hin, Hout, out, in, B, act, grid); break;
    case 4: launch_panel_inst<4>(st, W, bias, Hin, Hout, out, in, B, act, grid); break;
    case 5: launch_panel_inst<5>(st, W, bias, Hin, Hout, out, in, B, act, grid); break;
    case 6: launch_panel_inst<6>(st, W, bias, Hin, Hout, out, in, B, act, grid); break;
    case 7: launch_panel_inst<7>(st, W, bias, Hin, Hout, out, in, B, act, grid); break;
    default: launch_panel_inst<8>(st, W, bias, Hin, Hout, out, in, B, act, grid); break;
  }
  return true;
}

}  // namespace si
